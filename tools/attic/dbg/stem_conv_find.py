import sys, os, torch, time
sys.path.insert(0, os.getcwd())
for bench_flag in (False, True):
    torch.backends.cudnn.benchmark = bench_flag
    conv = torch.nn.Conv2d(3, 64, 7, stride=2, padding=3, bias=False).cuda()
    x = torch.randn(8, 3, 512, 512, device="cuda").requires_grad_()
    for it in range(3):
        y = conv(x); y.sum().backward()
    torch.cuda.synchronize()
    e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
    g = torch.randn_like(y)
    ts = []
    for it in range(10):
        e0.record(); y = conv(x); e1.record(); y.backward(g); e2.record(); torch.cuda.synchronize()
        ts.append((e0.elapsed_time(e1) * 1e3, e1.elapsed_time(e2) * 1e3))
    print("cudnn.benchmark", bench_flag, "fwd us %.1f  bwd us %.1f" % (min(t[0] for t in ts), min(t[1] for t in ts)))
