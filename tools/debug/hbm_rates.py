"""Developer tool (GPU box): what the memory system gives plain streaming kernels (torch fill / copy / read-reduce), to put the
codec kernels' byte rates in proportion."""
import torch, time
def t(fn, n=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
gb = 16
a = torch.empty(gb << 30, dtype=torch.uint8, device="cuda:0")
b = torch.empty(gb << 30, dtype=torch.uint8, device="cuda:0")
ai, bi = a.view(torch.int32), b.view(torch.int32)
ms = t(lambda: ai.fill_(7)); print("fill  %5.1f GB written           %.2f ms  %.2f TB/s" % (gb * 1.0737, ms, gb * 1.0737 / ms))
ms = t(lambda: bi.copy_(ai)); print("copy  %5.1f GB read + written    %.2f ms  %.2f TB/s (sum)" % (gb * 1.0737, ms, 2 * gb * 1.0737 / ms))
ms = t(lambda: ai.sum()); print("sum   %5.1f GB read              %.2f ms  %.2f TB/s" % (gb * 1.0737, ms, gb * 1.0737 / ms))
