import sys, os
sys.path.insert(0, os.getcwd())
import rtamd_loader
pkg = rtamd_loader.load()
t = pkg.RayTracer().create(64, 64, device_id=0)
t.render(0); t.synchronize()
print("tracer first: rendered, kernel ms", t.kernel_ms(0))
import torch
print("torch sees GPU:", torch.cuda.is_available(), torch.cuda.device_count())
x = torch.ones(4, device="cuda") * 2
print("torch result", x.sum().item())
t.render(1); t.synchronize(); print("ok")
