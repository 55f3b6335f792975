import os, sys, ctypes
print({k:v for k,v in os.environ.items() if 'HIP' in k or 'ROCR' in k or 'CUDA' in k or 'HSA' in k})
import torch
print('torch', torch.__version__, 'avail', torch.cuda.is_available(), 'count', torch.cuda.device_count())
try:
    x = torch.ones(4, device='cuda'); print('tensor ok', x.sum().item())
except Exception as e:
    print('tensor fail', repr(e)[:300])
sys.path.insert(0, '.')
import srt_amd
r = srt_amd.SoftwareRenderer(0); print('srt ctx ok')
try:
    y = torch.ones(4, device='cuda'); print('tensor after srt ok', y.sum().item())
except Exception as e:
    print('tensor after srt fail', repr(e)[:300])
with open('/proc/self/maps') as f:
    libs = sorted({l.split()[-1] for l in f if 'amdhip' in l or 'hsa-runtime' in l})
print(libs)
