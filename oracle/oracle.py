"""ctypes bindings for the CPU checkers -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
module (see oracle/wavenet_oracle.c).  The product package nv_wavenet_amd never does.

Two back-ends with the same Python surface:
  * ``Oracle``     -> oracle/liboracle.so, our plain-C restatement
                      (follows nv_wavenet_reference.cpp / matrix.cpp, cited in the C file)
  * ``RefOracle``  -> oracle/_ref/libnvwavenet_ref.so, the reference's own
                      nv_wavenet_reference.cpp + matrix.cpp compiled from /root/reference
                      (absent from git; built by ``make -C oracle ref`` where the reference
                      tree exists and shipped to the GPU box as a prebuilt file)
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_ORACLE_SO = os.path.join(_HERE, "liboracle.so")
_REF_SO = os.path.join(_HERE, "_ref", "libnvwavenet_ref.so")

_fp = C.POINTER(C.c_float)
_ip = C.POINTER(C.c_int)


def _f(a):
    assert a.dtype == np.float32 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(_fp)


def _i(a):
    assert a.dtype == np.int32 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(_ip)


def build(force=False):
    """Compile liboracle.so (and _ref when /root/reference is present)."""
    if force or not os.path.exists(_ORACLE_SO) or (
            os.path.getmtime(_ORACLE_SO) < os.path.getmtime(os.path.join(_HERE, "wavenet_oracle.c"))):
        subprocess.check_call(["make", "-C", _HERE, "oracle"], stdout=subprocess.DEVNULL)
    if os.path.exists("/root/reference/nv_wavenet_reference.cpp") and (force or not os.path.exists(_REF_SO)):
        subprocess.check_call(["make", "-C", _HERE, "ref"], stdout=subprocess.DEVNULL)


def have_ref():
    return os.path.exists(_REF_SO)


_libs = {}


def _lib(which):
    if which in _libs:
        return _libs[which]
    if which == "oracle":
        build()
        lib = C.CDLL(_ORACLE_SO)
        pfx = "nvw_oracle_"
        lib.nvw_oracle_create.restype = C.c_void_p
        lib.nvw_crc32.restype = C.c_uint32
        lib.nvw_crc32.argtypes = [C.c_void_p, C.c_size_t, C.c_uint32]
        gen = lib.nvw_gen_test_inputs
        lib._srand = lib.nvw_srand
    else:
        lib = C.CDLL(_REF_SO)
        pfx = "nvwref_"
        lib.nvwref_create.restype = C.c_void_p
        gen = lib.nvwref_gen_test_inputs
        lib._srand = lib.nvwref_srand
    lib._pfx = pfx
    lib._gen = gen
    gen.argtypes = [C.c_int] * 6 + [_fp] * 15
    gen.restype = None
    _libs[which] = lib
    return lib


class TestInputs:
    """One runTest() worth of inputs (nv_wavenet_test.cu:44-111,217-219), col-major weights."""
    __test__ = False  # not a pytest class

    def __init__(self, R, S, A, L, B, N):
        self.R, self.S, self.A, self.L, self.B, self.N = R, S, A, L, B, N
        z = lambda *s: np.zeros(s, dtype=np.float32)
        self.sel = z(N, B)
        self.embP, self.embC = z(A, R), z(A, R)
        self.Wprev, self.Wcur, self.Bh = z(L, R, 2 * R), z(L, R, 2 * R), z(L, 2 * R)
        self.Wres, self.Bres = z(L, R, R), z(L, R)
        self.Wskip, self.Bskip = z(L, R, S), z(L, S)
        self.Wzs, self.Bzs, self.Wza, self.Bza = z(S, A), z(A), z(A, A), z(A)
        self.Lh = z(N, L, B, 2 * R)

    def arrays(self):
        return [self.sel, self.embP, self.embC, self.Wprev, self.Wcur, self.Bh, self.Wres, self.Bres,
                self.Wskip, self.Bskip, self.Wzs, self.Bzs, self.Wza, self.Bza, self.Lh]

    def crc(self):
        lib = _lib("oracle")
        c = 0
        for a in self.arrays():
            c = lib.nvw_crc32(a.ctypes.data, a.nbytes, c)
        return c

    def round_to_half(self):
        """Round every weight / bias / embedding / conditioning value through IEEE fp16 (RNE),
        i.e. what an fp16 engine stores (nv_wavenet_conversions.cuh:28-36). Selectors stay fp32."""
        for a in self.arrays()[1:]:
            a[...] = a.astype(np.float16).astype(np.float32)
        return self


def gen_test_inputs(seed, prior, shape, which="oracle"):
    """srand(seed), generate (and drop) one runTest() input set per shape in ``prior``, then
    generate and return the set for ``shape`` = (R, S, A, L, B, N, ...): the reference runs
    several runTest() calls per srand() (nv_wavenet_test.cu:343-394), so later invocations see
    the rand() stream where the earlier ones left it."""
    lib = _lib(which)
    lib._srand(C.c_uint(seed))
    t = None
    for shp in list(prior) + [shape]:
        R, S, A, L, B, N = shp[:6]
        t = TestInputs(R, S, A, L, B, N)
        lib._gen(R, S, A, L, B, N, *[_f(a) for a in t.arrays()])
    return t


class _Base:
    which = None

    def __init__(self, L, maxBatch, maxSamples, R, S, A, maxDilation):
        self.lib = _lib(self.which)
        self.L, self.B, self.N, self.R, self.S, self.A = L, maxBatch, maxSamples, R, S, A
        self.maxDilation = maxDilation
        self._fn = lambda name: getattr(self.lib, self.lib._pfx + name)
        self.h = C.c_void_p(self._fn("create")(L, maxBatch, maxSamples, R, S, A, maxDilation))
        self._keep = []

    def close(self):
        if self.h:
            self._fn("destroy")(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_model(self, t):
        self._fn("set_embeddings")(self.h, _f(t.embP), _f(t.embC))
        for l in range(self.L):
            self._fn("set_layer_weights")(self.h, l, _f(t.Wprev[l]), _f(t.Wcur[l]), _f(t.Bh[l]),
                                          _f(t.Wres[l]), _f(t.Bres[l]), _f(t.Wskip[l]), _f(t.Bskip[l]))
        self._fn("set_out_weights")(self.h, _f(t.Wzs), _f(t.Bzs), _f(t.Wza), _f(t.Bza))

    def getters(self):
        """Last sample's activations in the engine's getter layouts:
        Xout [L][B][R], skipOut [L][B][S], Zs/Za/P [B][A]."""
        z = lambda *s: np.zeros(s, dtype=np.float32)
        X, K = z(self.L, self.B, self.R), z(self.L, self.B, self.S)
        for l in range(self.L):
            self._fn("get_xt_out")(self.h, l, _f(X[l]))
            self._fn("get_skip_out")(self.h, l, _f(K[l]))
        Zs, Za, P = z(self.B, self.A), z(self.B, self.A), z(self.B, self.A)
        self._fn("get_zs")(self.h, _f(Zs))
        self._fn("get_za")(self.h, _f(Za))
        self._fn("get_p")(self.h, _f(P))
        return dict(Xout=X, skipOut=K, Zs=Zs, Za=Za, P=P)


class Oracle(_Base):
    which = "oracle"

    def set_tanh_embed(self, flag):
        self.lib.nvw_oracle_set_tanh_embed(self.h, int(bool(flag)))

    def set_inputs(self, Lh, sel, copy=True):
        if not copy:
            self._keep = [Lh]
        self.lib.nvw_oracle_set_inputs(self.h, _f(Lh), _f(sel), int(copy))

    def run(self, num_samples, batch_size=None, forced=None, edges=False):
        """Returns yOut [B][num_samples] (and CDF edges lo, hi when edges=True)."""
        B = self.B if batch_size is None else batch_size
        y = np.zeros((self.B, num_samples), dtype=np.int32)
        lo = np.zeros((self.B, num_samples), dtype=np.float32) if edges else None
        hi = np.zeros((self.B, num_samples), dtype=np.float32) if edges else None
        rc = self.lib.nvw_oracle_run_ex(self.h, num_samples, B, _i(y),
                                        _i(forced) if forced is not None else None,
                                        _f(lo) if edges else None, _f(hi) if edges else None)
        if rc != 0:
            raise RuntimeError("oracle: selection fell off the CDF at sample %d "
                               "(the reference asserts, nv_wavenet_reference.cpp:119)" % (-rc - 1))
        return (y, lo, hi) if edges else y

    def history(self):
        p = np.zeros(self.B, dtype=np.int32)
        c = np.zeros(self.B, dtype=np.int32)
        self.lib.nvw_oracle_get_history(self.h, _i(p), _i(c))
        return p, c


class RefOracle(_Base):
    """The reference's own nvWavenetReference. batch_size must equal maxBatch (its matrix
    asserts require it, matrix.cpp:87-88) and it keeps every sample (RAM ~ N*B*L*R*4 B)."""
    which = "ref"

    def set_inputs(self, Lh, sel, copy=True):
        self.lib.nvwref_set_inputs(self.h, _f(Lh), _f(sel))

    def run(self, num_samples, batch_size=None):
        assert batch_size in (None, self.B)
        y = np.zeros((self.B, num_samples), dtype=np.int32)
        self.lib.nvwref_run(self.h, num_samples, self.B, _i(y))
        return y


def crc32(a, c=0):
    return _lib("oracle").nvw_crc32(a.ctypes.data, a.nbytes, c)


def philox4x32_10(ctr, key):
    """Random123 Philox4x32-10 of one (counter, key) pair: 4 uint32 words."""
    c = (C.c_uint32 * 4)(*ctr)
    k = (C.c_uint32 * 2)(*key)
    o = (C.c_uint32 * 4)()
    _lib("oracle").nvw_philox4x32_10(c, k, o)
    return [int(x) for x in o]


def philox_selectors(seed, N, B):
    """The [N][B] selector matrix the engine draws in-kernel for `seed` (wavenet_oracle.c)."""
    sel = np.zeros((N, B), dtype=np.float32)
    lib = _lib("oracle")
    lib.nvw_philox_selectors.argtypes = [C.c_uint64, C.c_int, C.c_int, _fp]
    lib.nvw_philox_selectors(int(seed), N, B, _f(sel))
    return sel


def mulaw_pcm_table(A):
    """int16 PCM value of every sample index (pytorch/utils.py:62-70 + inference.py:58-60)."""
    t = np.zeros(A, dtype=np.int16)
    _lib("oracle").nvw_mulaw_pcm_table(A, t.ctypes.data_as(C.c_void_p))
    return t
