"""What the vendor library (hipBLASLt / rocBLAS through torch.matmul) reaches on the multi-block GEMM shapes of this engine — a yardstick for
k_gemm_fat, not a product path (the package links no BLAS)."""
import torch, json
dev = 'cuda:0'
def t(M, K, N, reps=30):
    a = torch.randn(M, K, device=dev, dtype=torch.bfloat16); b = torch.randn(N, K, device=dev, dtype=torch.bfloat16)
    flush = torch.empty(1 << 28, dtype=torch.int32, device=dev)
    for _ in range(3): torch.matmul(a, b.t())
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        flush.add_(1)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); c = torch.matmul(a, b.t()); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    ts.sort()
    us = ts[len(ts) // 2]
    return us, 2.0 * M * K * N / us / 1e6
out = {}
for name, (K, N) in {'7b/mistral qkv(mistral)': (4096, 6144), '7b qkv': (4096, 12288), 'o_proj': (4096, 4096), 'gate/up mistral': (4096, 28672), 'down mistral': (14336, 4096),
                     'gate/up 7b': (4096, 22016), 'down 7b': (11008, 4096), '13b qkv': (5120, 15360), '13b o': (5120, 5120), '13b gate/up': (5120, 27648), '13b down': (13824, 5120),
                     'lm_head': (4096, 32000)}.items():
    for M in (256, 512):
        us, tf = t(M, K, N)
        out[f'{name} M={M}'] = (round(us, 1), round(tf, 1))
        print(f'{name:28s} M={M:4d} K={K:6d} N={N:6d}: {us:8.1f} us  {tf:7.1f} TFLOP/s')
print('RESULT', json.dumps(out))
