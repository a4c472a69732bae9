"""fills most of the device memory with a finite garbage pattern and exits (the next process then allocates dirty memory)"""
import sys, torch
dev = torch.device("cuda", 0)
val = float(sys.argv[1]) if len(sys.argv) > 1 else 3.0e4
bufs = [torch.full((1 << 28,), val, dtype=torch.float32, device=dev) for _ in range(200)]      # 200 GB
torch.cuda.synchronize()
print("filled", len(bufs), "GB with", val)
