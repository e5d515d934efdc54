"""dev: dump the compact PAR statistics of one library build: EXCEL_AB_LIB=... python tools_dev/par_stats_dump.py out.npy"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
if os.environ.get("EXCEL_AB_LIB"):
    import excel_amd._lib as _L
    _L.LIB_PATH = os.path.abspath(os.environ["EXCEL_AB_LIB"])
from excel_amd import ops
B, C, H, W = 2, 3, 64, 80
rs = np.random.RandomState(7)
img = torch.from_numpy(rs.standard_normal((B, 3, H, W)).astype(np.float32)).cuda()
masks = torch.from_numpy(rs.rand(B, C, H, W).astype(np.float32)).cuda()
need = ops.lib().excel_par_workspace_bytes(B, C, H, W, 6)
ws = torch.zeros(need, dtype=torch.uint8, device="cuda")
out = ops.par_forward(img, masks, (1, 2, 4, 8, 12, 24), 1, ws=ws)
torch.cuda.synchronize()
st = ws[:B * 5 * H * W * 4].view(torch.float32).reshape(B, 5, H, W).cpu().numpy()
np.save(sys.argv[1], st)
