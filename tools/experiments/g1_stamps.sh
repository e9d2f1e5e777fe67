cd ${GRAFT_REPO_ROOT:-/root/repo}
for shape in "32 512 512 0 1 2 1 0 8" "32 512 1536 0 1 2 1 0 8" "16 768 2304 0 1 2 1 0 8" "16 768 768 0 1 2 1 0 8"; do
  echo "== $shape"; timeout -k 5 60 build/ig_stamps $shape | tail -14
done
