"""care_absmax / the scaled split products against torch (debug probe; run on the GPU box)."""
import sys
import torch
sys.path.insert(0, ".")
from care_amd import training
from care_amd._lib import call, ptr

dev = "cuda"
for (M, K) in ((512, 320), (20992, 320), (87, 64), (3, 4)):
    for pos in range(8):
        x = torch.zeros(M, K, device=dev)
        x[M // 2, pos] = 3.0
        x[0, (pos + 1) % 4] = 1.0
        slot = torch.zeros(2, device=dev, dtype=torch.int32)
        call("care_absmax", ptr(x), K, M, K, slot.data_ptr())
        torch.cuda.synchronize()
        got = float(slot.view(torch.float32)[0])
        if got != 3.0:
            print("absmax M %d K %d max at column %d: got %r" % (M, K, pos, got))
x = torch.randn(20992, 320, device=dev)
slot = torch.zeros(2, device=dev, dtype=torch.int32)
call("care_absmax", ptr(x), 320, 20992, 320, slot.data_ptr())
torch.cuda.synchronize()
print("randn: kernel %r torch %r" % (float(slot.view(torch.float32)[0]), float(x.abs().max())))
for j in range(4):
    print("  max over elements with index %% 4 == %d: %r" % (j, float(x.view(-1, 4)[:, j].abs().max())))
