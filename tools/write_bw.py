import torch, time
x = torch.empty(2_000_000_000, dtype=torch.uint8, device="cuda")
y = torch.empty_like(x)
def t(f, n=20):
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
ms = t(lambda: x.fill_(7)); print("fill_ u8 2GB: %.3f ms -> %.2f TB/s" % (ms, 2e9/ms/1e9))
ms = t(lambda: x.zero_()); print("zero_ 2GB: %.3f ms -> %.2f TB/s" % (ms, 2e9/ms/1e9))
xf = x.view(torch.float32)
ms = t(lambda: xf.fill_(1.0)); print("fill_ f32 2GB: %.3f ms -> %.2f TB/s" % (ms, 2e9/ms/1e9))
ms = t(lambda: y.copy_(x)); print("copy 2GB: %.3f ms -> %.2f TB/s (r+w)" % (ms, 4e9/ms/1e9))
