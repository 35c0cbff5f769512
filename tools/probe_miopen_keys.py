import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["MIOPEN_USER_DB_PATH"] = os.path.join(ROOT, "miopen_db")
os.environ["MIOPEN_ENABLE_LOGGING_CMD"] = "1"
import torch
from transoar_amd.backbone import EncoderCnnBlock
torch.manual_seed(0)
blk = EncoderCnnBlock(48, 96, (3, 3, 3), (2, 2, 2)).cuda()
x = torch.randn(2, 48, 80, 80, 128, device="cuda", dtype=torch.bfloat16).contiguous(memory_format=torch.channels_last_3d).requires_grad_()
with torch.autocast("cuda", dtype=torch.bfloat16):
    y = blk(x)
    y.float().sum().backward()
torch.cuda.synchronize()
print("done", y.shape, y.stride())
