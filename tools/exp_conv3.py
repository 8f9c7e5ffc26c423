import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from svision_amd import kernels
dev = torch.device("cuda:0")
x = torch.randn(64, 256, 13, 13, device=dev).clamp_min(0); w = torch.randn(3, 3, 256, 384, device=dev) * 0.02
def timeit(fn, n=30, warm=5):
    for _ in range(warm): fn()
    torch.cuda.synchronize(); t=time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter()-t)/n*1e6
t = timeit(lambda: kernels.conv2d_same(x, w, None))
print("SVX_CONV_DEBUG=%s conv3: %.1f us  %.1f TF" % (os.environ.get("SVX_CONV_DEBUG","0"), t, 19.14/t*1e3))
