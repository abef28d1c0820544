"""Developer diagnostic (GPU box): which 512-entry chunks does the hinted write pass have to process?  Recomputes the kernel's
own skip test (binning2.hip: k_tile_pass<true>) from the scratch tables with torch."""
import ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sugar_amd import synthetic as syn, _lib
from sugar_amd.diff_gaussian_rasterization import _C, grad_sink
from tests import parity_utils as pu

scene, cams, bg = syn.make_config("metric")
cam = cams[0]
W, H = cam.image_width, cam.image_height
gx, gy = (W + 15) // 16, (H + 15) // 16
T = gx * gy
lib = C.CDLL(_lib.LIB_PATH)
dev = torch.device("cuda:0")
hint = torch.zeros(T, dtype=torch.int32, device=dev)
with grad_sink(tile_need_out=hint):
    hp = pu.run_hip(scene, cam, bg)
off = (C.c_size_t * 8)()
lib.sgr_debug_bin2_offsets(scene.means3D.shape[0], W, H, off)
img = _C.last_forward["img"].cpu().numpy()
T1, sgx, cap = int(off[4]), int(off[5]), int(off[6])
hdr = img[off[7]: off[7] + 32].view(np.uint32)
n_chunks = int(hdr[5])
sup_start = img[off[0]: off[0] + 4 * (T1 + 1)].view(np.uint32).astype(np.int64)
chunk_base = img[off[1]: off[1] + 4 * (T1 + 1)].view(np.uint32).astype(np.int64)
cnt2 = img[off[2]: off[2] + 4 * 64 * n_chunks].view(np.uint32).reshape(n_chunks, 64).astype(np.int64)
# chunk_info: one uint4 {super-tile, first level-1 entry, entries, -} per chunk (binning2.hip: k_sup_scan)
chunk_sup = img[off[3]: off[3] + 16 * n_chunks].view(np.uint32).reshape(n_chunks, 4)[:, 0].astype(np.int64)
need = hint.cpu().numpy().astype(np.int64)
ts = hp["tile_start"].astype(np.int64)
print("R1 =", hdr[4], "chunks =", n_chunks, "T1 =", T1)
lane = np.arange(64)
needed = np.zeros(n_chunks, bool)
why_small = 0
for c in range(n_chunks):
    sup = chunk_sup[c]
    tx = (sup % sgx) * 8 + (lane & 7); ty = (sup // sgx) * 8 + (lane >> 3)
    ok = (tx < gx) & (ty < gy)
    t = np.where(ok, ty * gx + tx, 0)
    nd = np.where(ok, need[t], 0)
    total = np.where(ok, ts[t + 1] - ts[t], 0)
    b = np.where(ok, cnt2[c], 0)
    m = b < nd
    needed[c] = m.any()
    if needed[c] and not (m & (b < total)).any():
        why_small += 1
print("chunks the kernel processes: %d of %d (%.3f)" % (needed.sum(), n_chunks, needed.mean()))
print("... of which only because a tile's WHOLE list is shorter than its hint: %d" % why_small)
