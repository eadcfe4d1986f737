"""Print the conv_pipe plan of a layer on the CPU (tuning build + DVF_PIPE_DEBUG): the C ABI is called with fake device
pointers; the plan is printed before the (failing, GPU-less) launch.  usage: plan_dump.py  (env DVF_PIPE_X4=0/1)"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["DVF_PIPE_DEBUG"] = "1"
lib = ctypes.CDLL(os.path.join(ROOT, "depth-vo-feat_amd/dvf/libdvf_hip_tuning.so"))
class Desc(ctypes.Structure):
    _fields_ = [(n, ctypes.c_int) for n in ("N", "C_in", "H_in", "W_in", "C_out", "H_out", "W_out", "KH", "KW", "stride", "pad", "transposed", "act")] + [("alpha", ctypes.c_float), ("beta", ctypes.c_float)]
LAYERS = [  # name, segs, cout, k, s, p, op, transposed, (N,H,W)
    ("iconv3 3x3s1 129->64 @64x208", [64, 64, 1], 64, 3, 1, 1, 0, 0, (4, 64, 208)),
    ("conv2.2 5x5s1 64->64 @64x208", [64], 64, 5, 1, 2, 0, 0, (4, 64, 208)),
    ("conv3.0 3x3s2 64->128", [64], 128, 3, 2, 1, 0, 0, (4, 64, 208)),
    ("conv3.2 3x3s1 128->128 @32x104", [128], 128, 3, 1, 1, 0, 0, (4, 32, 104)),
    ("pose up T4x4s2 128->64", [128], 64, 4, 2, 1, 0, 1, (4, 32, 104)),
    ("upconv3 T3x3s2 128->64", [128], 64, 3, 2, 1, 1, 1, (4, 32, 104)),
    ("upconv2 T3x3s2 64->32", [64], 32, 3, 2, 1, 1, 1, (4, 64, 208)),
    ("conv1.2 7x7s1 32->32", [32], 32, 7, 1, 3, 0, 0, (4, 128, 416)),
    ("iconv2 65->32 @128x416", [32, 32, 1], 32, 3, 1, 1, 0, 0, (4, 128, 416)),
    ("iconv4 256->128 @32x104", [128, 128], 128, 3, 1, 1, 0, 0, (4, 32, 104)),
    ("conv5.2 512->512 @8x26", [512], 512, 3, 1, 1, 0, 0, (4, 8, 26)),
    ("conv6.0 3x3s2 512->512 @8x26", [512], 512, 3, 2, 1, 0, 0, (4, 8, 26)),
    ("conv6.2 512->512 @4x13", [512], 512, 3, 1, 1, 0, 0, (4, 4, 13)),
    ("conv7.0 3x3s2 512->512 @4x13", [512], 512, 3, 2, 1, 0, 0, (4, 4, 13)),
    ("conv7.2 512->512 @2x7", [512], 512, 3, 1, 1, 0, 0, (4, 2, 7)),
    ("upconv7 T3x3s2 512->512 @2x7", [512], 512, 3, 2, 1, 1, 1, (4, 2, 7)),
    ("iconv7 1024->512 @4x13", [512, 512], 512, 3, 1, 1, 0, 0, (4, 4, 13)),
    ("upconv6 T3x3s2 512->512 @4x13", [512], 512, 3, 2, 1, 1, 1, (4, 4, 13)),
    ("pose conv6 3x3s2 256->256 @8x26", [256], 256, 3, 2, 1, 0, 0, (4, 8, 26)),
    ("pose conv7 3x3s2 256->256 @4x13", [256], 256, 3, 2, 1, 0, 0, (4, 4, 13)),
    ("head 128->1 @32x104", [128], 1, 3, 1, 1, 0, 0, (4, 32, 104)),
    ("head 128->2 @32x104", [128], 2, 3, 1, 1, 0, 0, (4, 32, 104)),
    ("head 64->1 @64x208", [64], 1, 3, 1, 1, 0, 0, (4, 64, 208)),
]
if len(sys.argv) > 1:
    LAYERS = [l for l in LAYERS if any(a in l[0] for a in sys.argv[1:])]
WSF = 1 << 26      # pretend split-K scratch (planning only)
for name, segs, cout, k, s, p, op, tr, (n, h, w) in LAYERS:
    if tr: oh, ow = (h - 1) * s - 2 * p + k + op, (w - 1) * s - 2 * p + k + op
    else: oh, ow = (h + 2 * p - k) // s + 1, (w + 2 * p - k) // s + 1
    d = Desc(n, sum(segs), h, w, cout, oh, ow, k, k, s, p, tr, 1, 1.0, 0.0)
    segc = (ctypes.c_int * len(segs))(*segs)
    fake = 0x7f0000000000
    ins = (ctypes.c_void_p * len(segs))(*[fake + 0x10000000 * (i + 1) for i in range(len(segs))])
    for kind, fn in ((0, "fwd"), (1, "dgrad")):
        print(f"== {name} {fn}", flush=True)
        sys.stdout.flush()
        if kind == 0:
            lib.dvf_conv2d_fwd_packed.restype = ctypes.c_int
            rc = lib.dvf_conv2d_fwd_packed(ctypes.byref(d), ins, segc, len(segs), ctypes.c_void_p(fake), ctypes.c_void_p(fake + 256), ctypes.c_void_p(fake + 4096), ctypes.c_void_p(fake + (1 << 34)), ctypes.c_int64(WSF), ctypes.c_void_p(0))
        else:
            rc = lib.dvf_conv2d_dgrad_packed(ctypes.byref(d), ctypes.c_void_p(fake), ctypes.c_void_p(fake + 4096), ctypes.c_void_p(fake + 8192), ins, segc, len(segs), ctypes.c_void_p(fake + (1 << 34)), ctypes.c_int64(WSF), ctypes.c_void_p(0))
