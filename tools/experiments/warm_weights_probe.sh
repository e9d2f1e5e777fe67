cd ${GRAFT_REPO_ROOT:-/root/repo}
for shape in "8 1024 1024 0 16 6 3 0" "8 2048 1024 0 16 6 3 0" "16 768 768 0 8 6 3 0" "16 1536 768 0 8 6 3 0" "32 512 512 0 2 6 3 0"; do
  for nb in 1 2 4 16; do echo -n "nbuf=$nb  "; timeout -k 5 60 build/ig_base $shape $nb 0 0 | tail -1; done
done
# 1x1 GEMMs: 16x16 qkv (768 -> 2304), 32x32 qkv (512 -> 1536)
for shape in "16 768 2304 0 1 2 1 0" "32 512 1536 0 1 2 1 0" "16 768 768 0 1 2 1 0"; do
  for nb in 1 16; do echo -n "nbuf=$nb  "; timeout -k 5 60 build/ig_base $shape $nb 0 0 | tail -1; done
done
