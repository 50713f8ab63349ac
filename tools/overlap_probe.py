"""Does a pinned-memory copy on one stream overlap a kernel stream on this box?  (diagnosis for bench.py's value_with_transfers)"""
import time, torch
dev = torch.device("cuda:0")
a = torch.randn(8192, 8192, device=dev, dtype=torch.bfloat16); b = torch.randn(8192, 8192, device=dev, dtype=torch.bfloat16)
h = torch.empty(64 << 20, dtype=torch.uint8, pin_memory=True); d = torch.empty(64 << 20, dtype=torch.uint8, device=dev)
h2 = torch.empty(64 << 20, dtype=torch.uint8, pin_memory=True); d2 = torch.empty(64 << 20, dtype=torch.uint8, device=dev)
sk, su, sd = torch.cuda.Stream(), torch.cuda.Stream(), torch.cuda.Stream()
def run(kern, up, down, n=20):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        if kern:
            with torch.cuda.stream(sk):
                for _ in range(4): torch.matmul(a, b)
        if up:
            with torch.cuda.stream(su): d.copy_(h, non_blocking=True)
        if down:
            with torch.cuda.stream(sd): h2.copy_(d2, non_blocking=True)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3
run(1, 1, 1, 3)
for k, u, dn in ((1, 0, 0), (0, 1, 0), (0, 0, 1), (0, 1, 1), (1, 1, 0), (1, 0, 1), (1, 1, 1)):
    print(f"kernels {k} upload {u} download {dn}: {run(k, u, dn):.3f} ms per iteration (4 GEMMs, 64 MiB each way)")

# the double-buffered structure of bench.py's measure_with_transfers, GEMMs as the step
nb = 2
up_done = [torch.cuda.Event() for _ in range(nb)]; comp_done = [torch.cuda.Event() for _ in range(nb)]; down_done = [torch.cuda.Event() for _ in range(nb)]
hb = [torch.empty(64 << 20, dtype=torch.uint8, pin_memory=True) for _ in range(nb)]; db = [torch.empty(64 << 20, dtype=torch.uint8, device=dev) for _ in range(nb)]
ho = [torch.empty(64 << 20, dtype=torch.uint8, pin_memory=True) for _ in range(nb)]; do = [torch.empty(64 << 20, dtype=torch.uint8, device=dev) for _ in range(nb)]
def upload(b):
    with torch.cuda.stream(su):
        su.wait_event(comp_done[b]); db[b].copy_(hb[b], non_blocking=True); up_done[b].record(su)
def download(b):
    with torch.cuda.stream(sd):
        sd.wait_event(comp_done[b]); ho[b].copy_(do[b], non_blocking=True); down_done[b].record(sd)
def pipeline(n, xfer):
    for b in range(nb): comp_done[b].record(sk)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    if xfer: upload(0)
    for i in range(n):
        b = i % nb
        if xfer:
            sk.wait_event(up_done[b])
            if i >= nb: sk.wait_event(down_done[b])
        with torch.cuda.stream(sk):
            for _ in range(4): torch.matmul(a, b_)
        comp_done[b].record(sk)
        if xfer:
            download(b)
            if i + 1 < n: upload((i + 1) % nb)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3
b_ = b
pipeline(4, True)
print(f"pipeline without transfers {pipeline(20, False):.3f} ms, with {pipeline(20, True):.3f} ms per step")

# the same pipeline with the step forked onto K streams and joined (bench.py's batch_step): does the copy overlap survive more streams than hardware queues?
def forked_step(K):
    base = torch.cuda.current_stream()
    for k in range(K):
        fs[k].wait_stream(base)
        with torch.cuda.stream(fs[k]):
            torch.matmul(a, b_)
    for k in range(K): base.wait_stream(fs[k])
fs = [torch.cuda.Stream() for _ in range(8)]
def pipeline_forked(n, xfer, K):
    for b in range(nb): comp_done[b].record(sk)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    if xfer: upload(0)
    for i in range(n):
        b = i % nb
        if xfer:
            sk.wait_event(up_done[b])
            if i >= nb: sk.wait_event(down_done[b])
        with torch.cuda.stream(sk):
            forked_step(K)
        comp_done[b].record(sk)
        if xfer:
            download(b)
            if i + 1 < n: upload((i + 1) % nb)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3
for K in (1, 2, 4, 8):
    pipeline_forked(4, True, K)
    print(f"forked onto {K} streams: without transfers {pipeline_forked(20, False, K):.3f} ms, with {pipeline_forked(20, True, K):.3f} ms per step")
