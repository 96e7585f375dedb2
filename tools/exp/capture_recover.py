"""Experiment: does a process survive a failed graph capture (a collective that cannot be captured)?"""
import os, torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
dist.init_process_group("gloo", rank=0, world_size=1)
dev = torch.device("cuda", 0)
x = torch.ones(1024, device=dev)
s = torch.cuda.Stream()
def body():
    y = x * 2
    dist.all_reduce(y)          # gloo on a CUDA tensor: host round trip, not capturable
    return y + 1
with torch.cuda.stream(s):
    body(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    try:
        with torch.cuda.graph(g, stream=s):
            body()
        print("captured?!")
    except Exception as e:
        print("capture failed:", type(e).__name__, str(e).splitlines()[0])
print("capturing now:", torch.cuda.is_current_stream_capturing())
for attempt in range(3):
    try:
        torch.cuda.synchronize()
        print("sync ok"); break
    except Exception as e:
        print("sync raised:", str(e).splitlines()[0])
s2 = torch.cuda.Stream()
try:
    with torch.cuda.stream(s2):
        print("eager after failure:", float(body().sum()))
    torch.cuda.synchronize()
    print("RECOVERED")
except Exception as e:
    print("eager failed:", str(e).splitlines()[0])
