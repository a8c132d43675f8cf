import time, torch, sys
sys.path.insert(0, '.')
from oracle import unet as ou
from sketch2img_amd import synthetic
from sketch2img_amd.config import SD15
W = synthetic.unet_state_dict(SD15)
x = torch.randn(2,4,64,64); ehs = torch.randn(2,77,768)
print("default threads", torch.get_num_threads(), flush=True)
for n in (32, 64, 128):
    torch.set_num_threads(n)
    t0=time.time()
    with torch.no_grad(): ou.unet_forward(ou.SD15, W, x, 981, ehs)
    print(n, "threads fwd", time.time()-t0, flush=True)
