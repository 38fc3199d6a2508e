"""NVWaveNet: the reference's Python wrapper class (/root/reference/pytorch/nv_wavenet.py:55-196)
on top of this repo's engine: same constructor keywords (what WaveNet.export_weights() returns,
pytorch/wavenet.py:147-188), same shape checks against the compiled R/S/A, same layout
conversion, same ``infer(cond_input, implementation)``.
"""
import torch

from . import nv_wavenet_ext
from .engine import Impl  # noqa: F401  (AUTO, SINGLE_BLOCK, DUAL_BLOCK, PERSISTENT, + MANYBLOCK)


def interleave_lists(*lists):
    return [x for t in zip(*lists) for x in t]


def column_major(x):
    """PyTorch tensors are row-major: return the contiguous transpose, i.e. the column-major image
    the engine expects (pytorch/nv_wavenet.py:33-49). 1-D unchanged; (M,K,1) squeezed; 4-D
    conditioning (2R,B,L,N) -> [N][L][B][2R]."""
    assert x.is_contiguous()
    if x.dim() == 1:
        return x
    if x.dim() == 3:
        assert x.size(2) == 1
        x = torch.squeeze(x, 2)
    if x.dim() == 2:
        return torch.t(x).contiguous()
    if x.dim() == 4:
        return x.permute(3, 2, 1, 0).contiguous()
    raise ValueError("unsupported tensor rank %d" % x.dim())


class NVWaveNet:
    def __init__(self, embedding_prev, embedding_curr, conv_out_weight, conv_end_weight, dilate_weights,
                 dilate_biases, max_dilation, res_weights, res_biases, skip_weights, skip_biases, use_embed_tanh):
        self.R, self.S, self.A = self._compiled_dims(embedding_prev, conv_out_weight)
        self.max_dilation = max_dilation
        self.use_embed_tanh = use_embed_tanh

        def check(name, t, shape):
            assert tuple(t.size()[:len(shape)]) == shape, \
                "%s: %s doesn't match compiled nv-wavenet size: %s" % (name, tuple(t.size()), shape)

        check("embedding_prev", embedding_prev, (self.A, self.R))
        check("embedding_curr", embedding_curr, (self.A, self.R))
        # the engine wants [A][R] row-major, which is what the tensors already are
        self.embedding_prev = column_major(torch.t(embedding_prev).contiguous())
        self.embedding_curr = column_major(torch.t(embedding_curr).contiguous())
        check("conv_out_weight", conv_out_weight, (self.A, self.S))
        self.conv_out = column_major(conv_out_weight.contiguous())
        check("conv_end_weight", conv_end_weight, (self.A, self.A))
        self.conv_end = column_major(conv_end_weight.contiguous())

        prev, curr = [], []
        for w in dilate_weights:
            assert w.size(2) == 2, "nv-wavenet only supports kernel_size 2"
            check("dilated weight", w, (2 * self.R, self.R))
            prev.append(column_major(w[:, :, 0].contiguous()))
            curr.append(column_major(w[:, :, 1].contiguous()))
        for b in dilate_biases:
            assert b.size(0) == 2 * self.R
        for w in res_weights:
            check("residual weight", w, (self.R, self.R))
        for b in res_biases:
            assert b.size(0) == self.R
        for w in skip_weights:
            check("skip weight", w, (self.S, self.R))
        for b in skip_biases:
            assert b.size(0) == self.S

        dilate_biases = [column_major(b.contiguous()) for b in dilate_biases]
        res_weights = [column_major(w.contiguous()) for w in res_weights]
        res_biases = [column_major(b.contiguous()) for b in res_biases]
        skip_weights = [column_major(w.contiguous()) for w in skip_weights]
        skip_biases = [column_major(b.contiguous()) for b in skip_biases]
        # the last layer's residual output is unused: pad zeros (pytorch/nv_wavenet.py:139-141)
        dev = embedding_prev.device
        res_weights.append(torch.zeros(self.R, self.R, device=dev))
        res_biases.append(torch.zeros(self.R, device=dev))
        assert len(res_biases) == len(skip_biases) == len(dilate_biases) and \
            len(res_weights) == len(skip_weights) == len(dilate_weights), \
            "Number of layers is inconsistent for different parameter types"
        self.num_layers = len(res_biases)
        self.layers = interleave_lists(prev, curr, dilate_biases, res_weights, res_biases, skip_weights,
                                       skip_biases)

    def _compiled_dims(self, embedding_prev, conv_out_weight):
        """(R, S, A) the tensors are checked against: the one instantiation behind wavenet_infer()."""
        return (nv_wavenet_ext.num_res_channels(), nv_wavenet_ext.num_skip_channels(),
                nv_wavenet_ext.num_out_channels())

    def infer(self, cond_input, implementation):
        """cond_input: channels(2R) x batch x num_layers x samples. Returns int32 [batch][samples]
        on cond_input's device (the reference returns a CUDA IntTensor, nv_wavenet.py:182)."""
        assert tuple(cond_input.size()[0:3:2]) == (2 * self.R, self.num_layers), \
            "Inputs are channels x batch x num_layers x samples; got %s" % (tuple(cond_input.size()),)
        batch_size = cond_input.size(1)
        sample_count = cond_input.size(3)
        cond_input = column_major(cond_input.contiguous())
        samples = torch.zeros(batch_size, sample_count, dtype=torch.int32, device=cond_input.device)
        nv_wavenet_ext.infer(samples, sample_count, batch_size, self.embedding_prev, self.embedding_curr,
                             self.conv_out, self.conv_end, cond_input, self.num_layers, self.use_embed_tanh,
                             self.max_dilation, implementation, self.layers)
        return samples


def cond_fragment_order(R, precision=16):
    """The engine's conditioning fragment order as a channel permutation: (perm, scale), both of length 2R.  Within one
    (sample, layer, tile of 16 utterances) the packed conditioning is [wave][fragment][g][utterance j][e]; position
    q = ((wave * fragments + fragment) * 4 + g) * EPL + e holds channel perm[q] of the reference's 2R gate channels (tanh rows
    0..R-1, sigmoid rows R..2R-1), multiplied by scale[q] -- the gate's pre-scale of the fp16 engine (2 log2 e on tanh rows,
    -log2 e on sigmoid rows: wn_kernels.hpp gate_prescale), 1 for fp32.  (Restates pack_cond_tiled_kernel's index arithmetic.)"""
    import math
    RT = R // 16
    NW = 4 if RT >= 4 else RT
    HTW = RT // NW
    TPF, EPL = (2, 8) if precision == 16 else (1, 4)
    CF = 2 * HTW // TPF
    perm, scale = [], []
    for w in range(NW):
        for c in range(CF):
            for g in range(4):
                for e in range(EPL):
                    it = c * TPF + (e >> 2)
                    tile16 = w + NW * (it >> 1) + (it & 1) * RT
                    ch = tile16 * 16 + g * 4 + (e & 3)
                    perm.append(ch)
                    scale.append(1.0 if precision != 16 else (-1.44269504088896340736 if ch >= R else 2.88539008177792681472))
    assert sorted(perm) == list(range(2 * R))
    return perm, scale


def _fragments_from_ordered(x, tiles, dtype):
    """x [N][L][B][2R] with the channels already in fragment order (and pre-scaled) -> the engine's packed tensor
    [N + 1][L][tiles][wave][fragment][g][j][EPL] (one zero padding sample, utterances padded to whole tiles)."""
    N, L, B, C2 = x.shape
    EPL = 8 if dtype == torch.float16 else 4
    buf = torch.zeros(N + 1, L, tiles * 16, C2, dtype=dtype, device=x.device)
    buf[:N, :, :B] = x.to(dtype)
    buf = buf.view(N + 1, L, tiles, 16, C2 // (4 * EPL), 4, EPL)          # [n][l][tile][j][wave*fragment][g][e]
    return buf.permute(0, 1, 2, 4, 5, 3, 6).contiguous()                    # [n][l][tile][wave*fragment][g][j][e]


def pack_cond_input(cond_nlbc, precision, tiles):
    """[samples][layers][batch][2R] conditioning (any float dtype, on the device) -> the engine's own fragment order, for
    WavenetEngine.setConditioningPacked / NVWaveNetEngine.infer(layout="packed"): a gather of the channel axis, the gate's
    pre-scale (fp16 engines), one cast, one permuting copy.  tiles = engine.condTiles()."""
    R = cond_nlbc.size(3) // 2
    perm, scale = cond_fragment_order(R, precision)
    dtype = torch.float16 if precision == 16 else torch.float32
    idx = torch.tensor(perm, device=cond_nlbc.device)
    x = cond_nlbc.float().index_select(3, idx)
    if precision == 16:
        x = x * torch.tensor(scale, dtype=torch.float32, device=x.device)
    return _fragments_from_ordered(x, tiles, dtype)


def cond_producer_weights(cond_weight, cond_bias, n_layers, precision=16):
    """The `cond_layers` 1x1 convolution (weight [2R*L][n_cond][1], bias [2R*L]) as the operands of the engine's fused producer
    (csrc/cond_producer.hip, nvw_produce_conditioning_f16): rows in the engine's fragment-position order with the gate's pre-scale
    folded in (cond_fragment_order), cut into MFMA A fragments.  Returns (wfrag fp16 [L][NWF][2][KF][64][8], bias fp32 [L][NWF*32],
    KF, NWF): fragment wf of a tile holds positions wf*32 .. wf*32+31; its row tile tt has row m = 4g + r at position
    (wf*4 + g)*8 + tt*4 + r, so that lane (g, j) of the two result tiles holds the 8 halves the packed layout gives it."""
    assert precision == 16, "the fused producer serves the fp16 engine"
    C2 = cond_weight.size(0) // n_layers
    n_cond = cond_weight.size(1)
    perm, scale = cond_fragment_order(C2 // 2, precision)
    dev = cond_weight.device
    idx = torch.tensor(perm, device=dev)
    sc = torch.tensor(scale, dtype=torch.float32, device=dev)
    w = cond_weight.reshape(n_layers, C2, n_cond).float().index_select(1, idx) * sc[None, :, None]          # rows = positions
    b = (cond_bias.reshape(n_layers, C2).float().index_select(1, idx) * sc[None, :]).contiguous()
    KF = (n_cond + 31) // 32
    NWF = C2 // 32
    w = torch.nn.functional.pad(w, (0, KF * 32 - n_cond))                                                     # [L][C2][32 KF]
    m = torch.arange(16, device=dev)
    rows = ((torch.arange(NWF, device=dev)[:, None, None] * 4 + (m // 4)[None, None, :]) * 8
            + torch.arange(2, device=dev)[None, :, None] * 4 + (m % 4)[None, None, :])                          # [wf][tt][i] -> position
    wg = w[:, rows.reshape(-1), :].reshape(n_layers, NWF, 2, 16, KF, 4, 8)                                     # [l][wf][tt][i][kf][ga][e]
    wfrag = wg.permute(0, 1, 2, 4, 5, 3, 6).contiguous().to(torch.float16)                                     # [l][wf][tt][kf][ga][i][e]
    return wfrag.reshape(n_layers, NWF, 2, KF, 64, 8), b, KF, NWF


_producer_cache = {}


def cond_producer_weights_cached(cond_weight, cond_bias, n_layers, precision=16):
    """cond_producer_weights, kept per weight tensor (storage + version): a streaming caller produces the conditioning once per
    chunk, and re-arranging the whole convolution weight every time cost as much as the producer kernel's own launch."""
    key = (cond_weight.data_ptr(), cond_weight._version, cond_bias.data_ptr(), cond_bias._version, n_layers, precision, tuple(cond_weight.shape))
    hit = _producer_cache.get(key)
    if hit is None:
        if len(_producer_cache) >= 4:
            _producer_cache.pop(next(iter(_producer_cache)))
        hit = _producer_cache[key] = cond_producer_weights(cond_weight, cond_bias, n_layers, precision)
    return hit


def produce_cond_packed(x_channels_last, wfrag, bias, out, tiles):
    """x [tiles*16][N][32 KF] fp16 (upsampled features, channels last, zero-padded) -> N samples of a packed conditioning buffer
    (out: [N][L][tiles][NWF][4][16][8] fp16, contiguous; e.g. a slice of what setConditioningPacked was given) on torch's
    current stream, by the engine's own kernel."""
    from ._lib import lib
    L, NWF, _, KF = wfrag.shape[:4]
    N = x_channels_last.size(1)
    assert x_channels_last.is_cuda and x_channels_last.dtype == torch.float16 and x_channels_last.is_contiguous()
    assert x_channels_last.shape == (tiles * 16, N, 32 * KF), (tuple(x_channels_last.shape), tiles, KF)
    assert out.is_cuda and out.dtype == torch.float16 and out.is_contiguous() and out.numel() == N * L * tiles * NWF * 512, tuple(out.shape)
    assert wfrag.is_contiguous() and bias.is_contiguous() and bias.dtype == torch.float32
    ok = lib.nvw_produce_conditioning_f16(x_channels_last.data_ptr(), wfrag.data_ptr(), bias.data_ptr(), out.data_ptr(), tiles, N, L, KF, NWF,
                                          torch.cuda.current_stream().cuda_stream)
    assert ok, "nvw_produce_conditioning_f16 refused its arguments"
    return out


def feature_fragments(x, tiles, precision=16):
    """Upsampled features x [B][n_cond][N] (any float dtype) -> the B-fragment order the generation kernels read
    (WavenetEngine.setConditioningFeatures): [N][tiles][KFC][4 g][16 j][EPL], fragment kf, lane (g, j), element e = channel
    (kf*TPF + (e>>2))*16 + 4g + (e&3) of utterance tile*16 + j, zero beyond n_cond / the batch.  (What pack_features_kernel writes.)"""
    B, Cn, N = x.shape
    TPF, EPL = (2, 8) if precision == 16 else (1, 4)
    KFC = -(-80 // (16 * TPF))
    KC = KFC * 16 * TPF
    assert Cn <= 80
    dtype = torch.float16 if precision == 16 else torch.float32
    buf = torch.zeros(tiles * 16, KC, N, dtype=dtype, device=x.device)
    buf[:B, :Cn] = x.to(dtype)
    buf = buf.view(tiles, 16, KFC, TPF, 4, 4, N)                      # [tile][j][kf][tk][g][r][n]
    return buf.permute(6, 0, 2, 4, 1, 3, 5).reshape(N, tiles, KFC, 4, 16, EPL).contiguous()


def upsample_features(features, upsample_weight, upsample_bias, upsample_stride, via_gemm=None):
    """The upsampling half of WaveNet.get_cond_input (pytorch/wavenet.py:195-197): ConvTranspose1d + trimming of its tail.
    Returns [B][n_cond][N] (a transposed view of the channels-last product on the GPU)."""
    import torch.nn.functional as F
    gemm = features.is_cuda if via_gemm is None else via_gemm
    if gemm and upsample_weight.size(2) % upsample_stride == 0:
        return _upsample_trimmed_gemm(features, upsample_weight, upsample_bias, upsample_stride).transpose(1, 2)
    x = F.conv_transpose1d(features, upsample_weight, upsample_bias, stride=upsample_stride)
    cutoff = upsample_weight.size(2) - upsample_stride
    return x[:, :, :-cutoff] if cutoff > 0 else x


def _upsample_trimmed_gemm(features, weight, bias, stride, pad_to=1):
    """ConvTranspose1d(kernel = m * stride, stride) followed by the trimming of its (kernel - stride) tail, as m matrix products
    (one per stride-long segment of the kernel, over the frames whose contribution survives the trimming):
    out[b][f*stride + r][co] = bias[co] + sum_j sum_ci features[b][ci][f - j] * weight[ci][co][j*stride + r], channels last, the
    channel dimension zero-padded to a multiple of `pad_to` (the fused producer reads whole 32-feature fragments).
    The library convolution compiles its kernels at first use on a machine without a kernel cache (MIOpen: a quarter of an hour
    of host time for bench.py's shapes on a fresh GPU box); a matrix product does not."""
    B, Ci, Fr = features.shape
    Co, K = weight.size(1), weight.size(2)
    m = K // stride
    Kp = (Ci + 31) // 32 * 32 if features.is_cuda else Ci           # (whole k-blocks for the GPU's matrix-product kernels)
    ft = features.transpose(1, 2)                                    # [B][f][ci]
    w = weight
    if Kp != Ci:
        ft = torch.nn.functional.pad(ft, (0, Kp - Ci))
        w = torch.nn.functional.pad(weight, (0, 0, 0, 0, 0, Kp - Ci))
    ft = ft.contiguous()
    acc = None                                                       # [B][f][co][r]
    for j in range(min(m, Fr)):
        y = torch.matmul(ft[:, :Fr - j], w[:, :, j * stride:(j + 1) * stride].reshape(Kp, Co * stride)).view(B, Fr - j, Co, stride)
        if acc is None:
            acc = y
        else:
            acc[:, j:] += y
    Cp = (Co + pad_to - 1) // pad_to * pad_to
    out = torch.zeros(B, Fr * stride, Cp, dtype=acc.dtype, device=acc.device) if Cp != Co else torch.empty(B, Fr * stride, Co, dtype=acc.dtype, device=acc.device)
    out.view(B, Fr, stride, Cp)[..., :Co] = acc.permute(0, 1, 3, 2)
    out[..., :Co] += bias
    return out


def get_cond_input(features, upsample_weight, upsample_bias, upsample_stride, cond_weight, cond_bias, n_layers,
                   layout="CBLN", dtype=None, precision=16, tiles=None, out=None, via_gemm=None, fused=None):
    """WaveNet.get_cond_input (pytorch/wavenet.py:190-202) as a function of the module's tensors,
    run wherever `features` lives (the GPU): ConvTranspose1d upsampling, trimming of the
    (kernel - stride) transposed-convolution tail, the 1x1 `cond_layers` convolution.
    features [B][n_cond][frames]; upsample_weight [n_cond][n_cond][kernel]; cond_weight
    [2R*n_layers][n_cond][1].  layout "CBLN" returns the reference's 2R x batch x layers x samples
    view; "NLBC" returns the engine's own [samples][layers][batch][2R] contiguous tensor, which
    NVWaveNetEngine.infer takes without a further permute/copy.  dtype=torch.float16 emits the conditioning in the fp16
    engine's T_data (rounded once, here), which that engine then reads in place at half the bytes.
    layout "packed" (round 3) emits the engine's own FRAGMENT order for an engine of `precision` bits whose condTiles() is
    `tiles` (NVWaveNetEngine.cond_tiles): the channel permutation and the gate's pre-scale are folded into the weights of the
    1x1 convolution -- its output channels simply come out in fragment order, scaled -- so the only extra work against "NLBC"
    is none: one permuting copy either way, and the generation kernels then run their packed path on the result as it is.
    fused: None = with layout="packed", out= and fp16 features on the GPU the conditioning convolution runs in the engine's own
    producer kernel (nvw_produce_conditioning_f16: fragment order written directly, no intermediate tensor); False forces the
    torch operations below.
    via_gemm: None = on the GPU the two convolutions run as matrix products (_upsample_trimmed_gemm; same sums in another order),
    on the CPU as the torch convolutions the reference's modules call (bit-identical to them); True / False force either."""
    import torch.nn.functional as F
    gemm = features.is_cuda if via_gemm is None else via_gemm
    can_fuse = (layout == "packed" and out is not None and precision == 16 and features.is_cuda and features.dtype == torch.float16
                and upsample_weight.size(2) % upsample_stride == 0 and cond_weight.size(1) <= 128)
    if (can_fuse if fused is None else fused):
        # the engine's own producer: upsampling as one matrix product (small), then the conditioning convolution by an MFMA kernel
        # that writes fragment order directly (csrc/cond_producer.hip)
        assert can_fuse, "fused=True needs layout='packed', out=, fp16 features on the GPU and a kernel that is a multiple of the stride"
        assert tiles is not None, "layout='packed' needs the engine's condTiles()"
        wfrag, bpos, KF, _ = cond_producer_weights_cached(cond_weight, cond_bias, n_layers, precision)
        xcl = _upsample_trimmed_gemm(features, upsample_weight, upsample_bias, upsample_stride, pad_to=32)   # [B][N][32 KF]
        assert xcl.size(0) == tiles * 16, "writing into a packed buffer needs whole tiles (pad the batch)"
        return produce_cond_packed(xcl, wfrag, bpos, out, tiles)
    if gemm and upsample_weight.size(2) % upsample_stride == 0:
        x = _upsample_trimmed_gemm(features, upsample_weight, upsample_bias, upsample_stride)         # [B][N][n_cond]
        # [B][N][n_cond] x [n_cond][channels] -> the [B][channels][N] the code below expects, as a view of the channels-last product
        conv1x1 = lambda t, w, b: torch.matmul(t, w.reshape(w.size(0), w.size(1)).t()).add_(b).transpose(1, 2)
    else:
        x = F.conv_transpose1d(features, upsample_weight, upsample_bias, stride=upsample_stride)
        cutoff = upsample_weight.size(2) - upsample_stride
        if cutoff > 0:
            x = x[:, :, :-cutoff]
        conv1x1 = F.conv1d
    if layout == "packed":
        assert tiles is not None, "layout='packed' needs the engine's condTiles()"
        C2 = cond_weight.size(0) // n_layers
        perm, scale = cond_fragment_order(C2 // 2, precision)
        idx = torch.tensor(perm, device=cond_weight.device)
        sc = torch.tensor(scale, dtype=cond_weight.dtype, device=cond_weight.device)
        w = cond_weight.view(n_layers, C2, cond_weight.size(1), 1).index_select(1, idx) * sc[None, :, None, None]
        b = cond_bias.view(n_layers, C2).index_select(1, idx) * sc[None, :]
        x = conv1x1(x, w.reshape(n_layers * C2, cond_weight.size(1), 1), b.reshape(-1))        # [B][L * 2R][N]
        if out is not None:
            # straight into N samples of an existing packed buffer: [tile][j][l][wave*fragment][g][e][n] -> [n][l][tile][wf][g][j][e]
            EPL = 8 if precision == 16 else 4
            assert x.size(0) == tiles * 16, "writing into a packed buffer needs whole tiles (pad the batch)"
            out.copy_(x.view(tiles, 16, n_layers, C2 // (4 * EPL), 4, EPL, x.size(2)).permute(6, 2, 0, 3, 4, 1, 5))
            return out
        x = x.view(x.size(0), n_layers, C2, x.size(2)).permute(3, 1, 0, 2)            # [N][L][B][2R], fragment channel order
        return _fragments_from_ordered(x, tiles, torch.float16 if precision == 16 else torch.float32)
    x = conv1x1(x, cond_weight, cond_bias)
    x = x.view(x.size(0), n_layers, -1, x.size(2))          # [B][L][2R][N]
    if layout == "CBLN":
        x = x.permute(2, 0, 1, 3)
        return x if dtype is None else x.to(dtype)
    assert layout == "NLBC"
    x = x.permute(3, 1, 0, 2)
    return (x if dtype is None else x.to(dtype)).contiguous()


class NVWaveNetEngine(NVWaveNet):
    """The same wrapper with the engine kept alive: the reference rebuilds
    its nvWavenetInfer object and re-uploads every weight on each infer() (wavenet_infer.cu:124-143).
    Here the weights are uploaded once per (batch, samples) engine, R/S/A are taken from the
    tensors (any instantiation in the build, fp32 or fp16), conditioning may already be in the
    engine's layout on the device, selectors can be drawn in-kernel from a seed, and int16 audio
    can be returned beside the indices."""

    def __init__(self, *args, precision=32, **kwargs):
        self.precision = precision
        self._engines = {}
        super().__init__(*args, **kwargs)

    def _compiled_dims(self, embedding_prev, conv_out_weight):
        from .engine import supported_configs
        dims = (embedding_prev.size(1), conv_out_weight.size(1), embedding_prev.size(0))
        assert dims + (self.precision,) in set(supported_configs()), \
            "no engine for R,S,A,precision = %s in this build" % (dims + (self.precision,),)
        return dims

    # engines are kept per (batch, sample CAPACITY, implementation): the capacity is the sample count rounded up
    # to a bucket, and an utterance shorter than the capacity runs as a prefix (conditioning, selectors and
    # samples are sample-major), so a new utterance length neither rebuilds the engine nor re-uploads the
    # weights.  At most MAX_ENGINES stay alive (least recently used goes first).
    BUCKET = 4096
    MAX_ENGINES = 4

    def _engine(self, batch_size, sample_count, implementation):
        from .engine import WavenetEngine
        capacity = -(-sample_count // self.BUCKET) * self.BUCKET
        key = (batch_size, capacity, int(implementation))
        e = self._engines.pop(key, None)
        if e is None:
            while len(self._engines) >= self.MAX_ENGINES:
                old_key = next(iter(self._engines))
                self._engines.pop(old_key).close()
            e = WavenetEngine(self.R, self.S, self.A, self.num_layers, self.max_dilation, batch_size, capacity,
                              impl=int(implementation), tanhEmbed=bool(self.use_embed_tanh), precision=self.precision)
            f = lambda t: t.float().contiguous()
            e.setEmbeddings(f(self.embedding_prev), f(self.embedding_curr))
            for l in range(self.num_layers):
                e.setLayerWeights(l, *[f(t) for t in self.layers[7 * l:7 * l + 7]])
            zeros = torch.zeros(self.A, dtype=torch.float32, device=self.conv_out.device)
            e.setOutWeights(f(self.conv_out), zeros, f(self.conv_end), zeros)   # wavenet_infer.cu:75-82
        self._engines[key] = e          # (re-)inserted last: most recently used
        return e

    def close(self):
        for e in self._engines.values():
            e.close()
        self._engines = {}

    def cond_tiles(self, batch_size, sample_count, implementation=Impl.AUTO):
        """condTiles() of the engine that infer() will use for this batch and utterance length (for layout "packed")."""
        return self._engine(batch_size, sample_count, implementation).condTiles()

    def infer(self, cond_input, implementation=Impl.AUTO, seed=None, return_audio=False, layout="CBLN",
              generator=None, batch_size=None):
        """cond_input: 2R x batch x layers x samples (layout "CBLN", the reference's: converted to the engine's fragment order
        by the one permuting copy any layout change needs), [samples][layers][batch][2R] contiguous (layout "NLBC", read in
        place), or the engine's own fragment order
        (layout "packed": get_cond_input(..., layout="packed") / pack_cond_input; batch_size must be given; the generation
        kernels run their packed path on it, no copy and no conversion).
        seed: int -> selectors drawn in-kernel by Philox4x32-10; None -> torch.rand on the device.
        Returns int32 [batch][samples] (and int16 audio [batch][samples] when return_audio)."""
        if layout == "packed":
            assert batch_size is not None and cond_input.is_cuda and cond_input.is_contiguous()
            return self._infer_packed(cond_input, batch_size, implementation, seed, return_audio, generator)
        if layout == "CBLN":
            # the reference's layout: one permuting copy is needed whatever the engine reads, so it goes straight to the
            # engine's fragment order (pack_cond_input) and the generation kernels run their packed path on it
            assert tuple(cond_input.size()[0:3:2]) == (2 * self.R, self.num_layers), \
                "Inputs are channels x batch x num_layers x samples; got %s" % (tuple(cond_input.size()),)
            dev = self.conv_out.device
            sample_count, batch_size = cond_input.size(3), cond_input.size(1)
            e = self._engine(batch_size, sample_count, implementation)
            frags = pack_cond_input(cond_input.to(device=dev).permute(3, 2, 1, 0), self.precision, e.condTiles())
            stream = torch.cuda.current_stream(dev)
            stream.synchronize()
            e.setConditioningPacked(frags, sample_count)
            return self._generate(e, dev, stream, sample_count, batch_size, seed, return_audio, generator)
        assert layout == "NLBC" and cond_input.is_contiguous() and tuple(cond_input.size()[1::2]) == (self.num_layers, 2 * self.R)
        sample_count, batch_size = cond_input.size(0), cond_input.size(2)
        # the conditioning lives where the model lives (a host tensor is uploaded once, like the reference's setInputs
        # does); an fp16 engine reads an fp16 tensor as it is, everything else is consumed as fp32
        dev = self.conv_out.device
        keep_half = self.precision == 16 and cond_input.dtype == torch.float16
        cond_input = cond_input.to(device=dev, dtype=torch.float16 if keep_half else torch.float32).contiguous()
        e = self._engine(batch_size, sample_count, implementation)
        # everything below is ordered on the caller's current stream of that device: the tensors above were produced
        # on it, the engine's launches go to it, and the results are consumed on it
        stream = torch.cuda.current_stream(dev)
        stream.synchronize()                # (the engine's own uploads run on its upload stream)
        # consumed in place from this tensor: no packed copy; it stays referenced until the run below has completed
        e.setConditioningDirect(cond_input, sample_count)
        return self._generate(e, dev, stream, sample_count, batch_size, seed, return_audio, generator)

    def infer_features(self, x, cond_weight, cond_bias, implementation=Impl.AUTO, seed=None, return_audio=False, generator=None):
        """Generation with the conditioning convolution INSIDE the kernel (round 5): x = the upsampled features [batch][n_cond][samples]
        on the GPU (upsample_features(...) / the model's self.upsample output, trimmed), cond_weight / cond_bias = the model's
        cond_layers.weight / .bias.  Equivalent to infer(model.get_cond_input(features)) without ever building the
        2R x batch x layers x samples tensor."""
        batch_size, sample_count = x.size(0), x.size(2)
        e = self._engine(batch_size, sample_count, implementation)
        dev = x.device
        key = (cond_weight.data_ptr(), cond_weight._version, cond_bias.data_ptr(), cond_bias._version)
        if getattr(e, "_cond_w_key", None) != key:
            e.setConditioningWeights(cond_weight.float().contiguous(), cond_bias.float().contiguous())
            e._cond_w_key = key
        stream = torch.cuda.current_stream(dev)
        stream.synchronize()
        e.setFeatures(x, sample_count)
        return self._generate(e, dev, stream, sample_count, batch_size, seed, return_audio, generator)

    def _infer_packed(self, frags, batch_size, implementation, seed, return_audio, generator):
        sample_count = frags.size(0) - 1
        e = self._engine(batch_size, sample_count, implementation)
        assert frags.size(2) == e.condTiles(), "packed conditioning for %d tiles, the engine expects %d" % (frags.size(2), e.condTiles())
        dev = frags.device
        stream = torch.cuda.current_stream(dev)
        stream.synchronize()
        e.setConditioningPacked(frags, sample_count)
        return self._generate(e, dev, stream, sample_count, batch_size, seed, return_audio, generator)

    def _generate(self, e, dev, stream, sample_count, batch_size, seed, return_audio, generator):
        sptr = stream.cuda_stream
        if seed is None:
            sel = torch.rand(sample_count, batch_size, dtype=torch.float32, device=dev, generator=generator)
            stream.synchronize()
            e.setSelectors(sel, sample_count)
        else:
            e.setSelectorSeed(seed)
        samples = torch.zeros(batch_size, sample_count, dtype=torch.int32, device=dev)
        audio = torch.zeros(batch_size, sample_count, dtype=torch.int16, device=dev) \
            if return_audio else None
        e.setAudioOut(audio)
        bspb = 4 if batch_size % 4 == 0 else 2 if batch_size % 2 == 0 else 1
        ok = e.run(sample_count, batch_size, samples, bspb, False, sptr)
        stream.synchronize()
        e.setAudioOut(None)
        assert ok and e.chainStatus() == 0, "nvWavenetInfer::run failed"   # (a multi-CU launch that gave up was re-run on wavenet_wg)
        return (samples, audio) if return_audio else samples
