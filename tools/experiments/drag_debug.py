"""debug aid: zero-displacement drag loss at the full tap size"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from ishapediting_amd import synthetic
from ishapediting_amd.drag_utils import DragKernels, feat_channel_map
dev = torch.device("cuda:0")
ch, width = 512, 64
feat = (torch.randn((width * width, ch), generator=torch.Generator().manual_seed(11))).half().to(dev)
src, _ = synthetic.handles(3, seed=7)
for cof in (0.0, 0.4):
    dk = DragKernels(dev, W=width, ld=ch, chmap=feat_channel_map(ch), r=12, voxel=2.0 / 256, loss_type="l2")
    dk.setup(src, src, cof)
    for rep in range(2):
        grad, loss = dk.loss_grad(feat, feat.clone())
        torch.cuda.synchronize()
        print("cof", cof, "rep", rep, "loss", float(loss), "max|grad|", float(grad.abs().max()), "nonzero grads", int((grad != 0).sum()),
              "acc", dk.acc.tolist(), "gfx max", int(dk.gfx.abs().max()))
