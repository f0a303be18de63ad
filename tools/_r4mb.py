import os, sys, time, torch
sys.path.insert(0, '.')
def t(fn, k=40):
    for _ in range(8): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(k): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / k * 1e6
tun = torch.cuda.tunable
tun.set_filename("gpurun_out/r4j/tune_mb.csv", insert_device_ordinal=False)
tun.set_max_tuning_duration(30); tun.set_max_tuning_iterations(30)
tun.enable(True); tun.tuning_enable(True)
M = 24576
dev = "cuda"
for (n, k, S) in [(512, 348, 8), (256, 512, 8), (128, 256, 16)]:
    dz = torch.randn(2, M, n, device=dev); x = torch.randn(2, M, k, device=dev)
    one = t(lambda: [torch.bmm(dz[i].view(S, M // S, n).transpose(1, 2), x[i].view(S, M // S, k)) for i in range(2)])
    res = [f"2 x bmm/{S}: {one:7.1f}"]
    for S2 in (S, S // 2, S * 2):
        a = dz.view(2 * S2, M // S2, n); b = x.view(2 * S2, M // S2, k)
        both = t(lambda: torch.bmm(a.transpose(1, 2), b))
        res.append(f"bmm/{2*S2} joint: {both:7.1f}")
    print(f"dW n={n} k={k}: " + " | ".join(res), flush=True)
for (n, k) in [(256, 512), (128, 256)]:
    dz = torch.randn(2, M, n, device=dev); w = torch.randn(2, n, k, device=dev)
    two = t(lambda: [dz[i] @ w[i] for i in range(2)])
    joint = t(lambda: torch.bmm(dz, w))
    print(f"dX n={n} k={k}: 2 x mm {two:7.1f} | bmm/2 {joint:7.1f}", flush=True)
# gathers: once per update vs per step
B = 4 * M
obs = torch.randn(B, 348, device=dev); perm = torch.randperm(B, device=dev)
print("gather M rows:", t(lambda: obs[perm[:M]]), " gather all 4M rows:", t(lambda: obs[perm]))
