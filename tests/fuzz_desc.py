"""Random descriptors and kernel-selection codes against the C ABI on a machine WITHOUT a GPU (tests/test_abi.py runs it in a subprocess)."""
import ctypes as C, os, sys, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from anakin_amd import lib as L
lib = L.load()
random.seed(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
EDGE = [-2147483648, -7, -1, 0, 1, 2, 3, 4, 5, 7, 8, 15, 16, 17, 31, 32, 33, 63, 64, 65, 127, 128, 255, 256, 1000, 1 << 16, 1 << 20, (1 << 31) - 1]
SMALL = [0, 1, 2, 3, 4, 7, 8, 14, 16, 28, 56, 64, 112, 224, 256]
def val(): return random.choice(EDGE if random.random() < 0.35 else SMALL)
codes = {}
for it in range(int(sys.argv[2]) if len(sys.argv) > 2 else 3000):
    d = L.ConvDesc()
    for name, ty in d._fields_:
        if ty is C.c_int:
            setattr(d, name, val())
        else:
            setattr(d, name, random.choice([0.0, 1.0, -1.0, 1e-30, 1e30, float("inf"), float("nan"), 0.5]))
    if random.random() < 0.7:      # mostly plausible enums so that deeper validation is reached
        d.in_dtype, d.out_dtype = random.choice([0, 1, 2]), random.choice([0, 1, 2])
        d.in_layout, d.out_layout = random.choice([0, 1]), random.choice([0, 1])
        d.int8_weights = random.choice([0, 1]); d.act = random.choice([0, 1, 2, 3]); d.res_mode = random.choice([0, 1, 2, 3]); d.group = random.choice([1, 1, 1, 2, 4, 0])
        d.stride_h = d.stride_w = random.choice([1, 2, 0, 3]); d.dil_h = d.dil_w = random.choice([1, 1, 2, 0])
    h = C.c_void_p()
    rc = lib.saber_hip_conv2d_create(C.byref(d), C.byref(h))
    codes[rc] = codes.get(rc, 0) + 1
    assert rc in (0, -1, -2, -3, -4, -5, -6, 1, 2, 3, 4, 5), rc
    if rc == 0:
        oh, ow = C.c_int(), C.c_int()
        lib.saber_hip_conv2d_get_tile(h)
        lib.saber_hip_conv2d_destroy(h)
    f = L.FcDesc()
    for name, ty in f._fields_:
        setattr(f, name, val())
    hf = C.c_void_p()
    rc = lib.saber_hip_fc_create(C.byref(f), C.byref(hf))
    codes[("fc", rc)] = codes.get(("fc", rc), 0) + 1
    if rc == 0:
        lib.saber_hip_fc_destroy(hf)
    lib.saber_hip_pool_out_dim(val(), val(), val(), val(), random.choice([0, 1]))
print("ok", codes)
# ---- plausible operators: create succeeds, then random kernel-selection codes (saber_hip_conv2d_set_tile) - a status, never a crash ----
ok = bad = st_ok = st_bad = 0
for it in range(2000):
    d = L.ConvDesc()
    d.n, d.h, d.w = random.choice([1, 2, 8]), random.choice([1, 7, 14, 28, 56, 224]), random.choice([1, 7, 14, 28, 56, 224])
    d.c, d.k = random.choice([3, 4, 16, 32, 64, 96, 128, 256, 512, 2048]), random.choice([8, 16, 40, 64, 128, 256, 1000, 2048])
    d.kh = d.kw = random.choice([1, 1, 3, 3, 7, 5])
    d.pad_h = d.pad_w = random.choice([0, 1, 3])
    d.stride_h = d.stride_w = random.choice([1, 1, 2]); d.dil_h = d.dil_w = random.choice([1, 1, 2]); d.group = 1
    i8 = random.random() < 0.5
    d.int8_weights = int(i8)
    d.in_dtype = random.choice([L.S8, L.U8, L.F32]) if i8 else L.F32
    d.out_dtype = random.choice([L.S8, L.U8, L.F32]) if i8 else L.F32
    d.in_layout = L.NHWC if i8 and d.in_dtype != L.F32 else random.choice([L.NHWC, L.NCHW])
    d.out_layout = L.NHWC if i8 and d.out_dtype != L.F32 else random.choice([L.NHWC, L.NCHW])
    d.act = random.choice([0, 1]); d.res_mode = random.choice([0, 0, 1, 2, 3])
    d.sum_scale = d.coeff_conv = d.coeff_res = d.scale_res = 1.0
    h = C.c_void_p()
    rc = lib.saber_hip_conv2d_create(C.byref(d), C.byref(h))
    if rc != 0:
        bad += 1
        continue
    ok += 1
    for _ in range(12):
        code = random.choice([random.getrandbits(24), random.getrandbits(8) | (random.choice([0, 1, 2, 4, 0x11, 0x21, 0x31, 0xff]) << 8) | (random.choice(range(0, 17)) << 16), random.getrandbits(31)])
        r2 = lib.saber_hip_conv2d_set_tile(h, code)
        st_ok += r2 == 0
        st_bad += r2 != 0
        lib.saber_hip_conv2d_get_tile(h)
    lib.saber_hip_conv2d_destroy(h)
print("plausible convs: created %d, refused %d; set_tile accepted %d, refused %d" % (ok, bad, st_ok, st_bad))
