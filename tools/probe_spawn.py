"""Probe (GPU box): may a process that has initialised the GPU start child processes (subprocess / multiprocessing
spawn)?  The two-rank data-parallel GPU test depends on the answer."""
import subprocess
import sys

import torch

torch.zeros(1, device="cuda").sum().item()
print("parent initialised the GPU", flush=True)
r = subprocess.run([sys.executable, "-c", "import torch; print('child sees', torch.cuda.device_count(), 'gpu;', float(torch.ones(2, device='cuda').sum()))"],
                   capture_output=True, text=True, timeout=300)
print("subprocess rc", r.returncode, r.stdout.strip(), r.stderr.strip()[-300:], flush=True)
