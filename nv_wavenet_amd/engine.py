"""WavenetEngine: the nvWavenetInfer<T_weight,T_data,R,S,A> class surface
(/root/reference/nv_wavenet.cuh:220-640) reached through the C ABI (include/nv_wavenet_c.h).

Method names, argument order, defaults and layouts are the reference's; arrays may be numpy
(host) or torch (host or device) -- the engine copies them, like the reference does.
"""
import ctypes as C

import numpy as np

from ._lib import lib, addr, CONSUME_FN


class Impl:
    """nvWavenetInfer::Implementation (nv_wavenet.cuh:223-229). The reference's Python enum
    (pytorch/nv_wavenet.py:53) stops at PERSISTENT; MANYBLOCK is exposed here as well."""
    AUTO = 0
    SINGLE_BLOCK = 1
    DUAL_BLOCK = 2
    PERSISTENT = 3
    MANYBLOCK = 4


class Org:
    """Kernel organisations (nvwOrganisation in nv_wavenet.hpp; last argument of nvw_create_ex)."""
    AUTO = 0        # from the Implementation value and the batch size
    WG = 1          # wn::wavenet_wg, 1, 2 or 3 tiles of 16 utterances per workgroup by batch size
    WG1 = 2
    WG2 = 3
    WG3 = 4         # three tiles per workgroup (fp16, R <= 64; else two)
    CHAIN = 5       # wn::wavenet_chain: multi-CU, weights resident, fewest CUs
    CHAIN1 = 6      # wn::wavenet_chain, one layer per CU
    # (7, 8, 9 were wn::wavenet_bcast and its variants, removed in round 5: refused by nvw_create_ex)
    WG4 = 10        # four tiles per workgroup (round 6: fp16, R <= 64, dump-free launches; else three)
    BY_NAME = {None: 0, "auto": 0, "wg": 1, "wg1": 2, "wg2": 3, "wg3": 4, "chain": 5, "chain1": 6, "wg4": 10}


def supported_configs():
    buf = (C.c_int * 256)()
    n = lib.nvw_list_supported(buf, 64)
    return [tuple(buf[4 * i:4 * i + 4]) for i in range(min(n, 64))]


def _f32(x):
    if hasattr(x, "data_ptr"):
        import torch
        assert x.dtype == torch.float32, "expected float32"
        return x
    x = np.ascontiguousarray(x, dtype=np.float32)
    return x


class WavenetEngine:
    def __init__(self, R, S, A, numLayers, maxDilation, batchSize, numSamples, impl=0, tanhEmbed=True,
                 precision=32, organisation=0):
        if not lib.nvw_supported(R, S, A, precision):
            raise ValueError("no nvWavenetInfer<%s,R=%d,S=%d,A=%d> in this build; have %s" %
                             ("half2,half" if precision == 16 else "float,float", R, S, A, supported_configs()))
        if impl not in (0, 1, 2, 3, 4):
            raise ValueError("implementation must be 0..4")
        self.R, self.S, self.A = R, S, A
        self.numLayers, self.maxDilation = numLayers, maxDilation
        self.maxBatch, self.maxSamples = batchSize, numSamples
        self.precision = precision
        if isinstance(organisation, str) or organisation is None:
            organisation = Org.BY_NAME[organisation]
        if organisation not in set(Org.BY_NAME.values()):
            raise ValueError("organisation must be one of %s" % sorted(set(Org.BY_NAME.values())))
        self._h = lib.nvw_create_ex(R, S, A, precision, numLayers, maxDilation, batchSize, numSamples, impl,
                                    1 if tanhEmbed else 0, organisation)
        if not self._h:
            raise RuntimeError("nvw_create failed: shape not supported in this organisation")
        self._cb_keep = None

    def close(self):
        if getattr(self, "_h", None):
            lib.nvw_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- model ------------------------------------------------------------------------------
    def setEmbeddings(self, embedPrev, embedCur):
        p, c = _f32(embedPrev), _f32(embedCur)
        lib.nvw_set_embeddings(self._h, addr(p), addr(c))

    def setLayerWeights(self, layer, Wprev, Wcur, Bh, Wres, Bres, Wskip, Bskip):
        a = [_f32(x) for x in (Wprev, Wcur, Bh, Wres, Bres, Wskip, Bskip)]
        lib.nvw_set_layer_weights(self._h, layer, *[addr(x) for x in a])

    def setOutWeights(self, Wzs, Bzs, Wza, Bza):
        a = [_f32(x) for x in (Wzs, Bzs, Wza, Bza)]
        lib.nvw_set_out_weights(self._h, *[addr(x) for x in a])

    def setInputs(self, Lh, outputSelectors, numSamples=None):
        """Lh [numSamples][L][maxBatch][2R] fp32, outputSelectors [numSamples][maxBatch] fp32;
        numSamples defaults to the engine's capacity (the reference's setInputs), may be smaller."""
        Lh, sel = _f32(Lh), _f32(outputSelectors)
        ns = self.maxSamples if numSamples is None else int(numSamples)
        assert 0 < ns <= self.maxSamples
        n = ns * self.numLayers * self.maxBatch * 2 * self.R
        assert (Lh.numel() if hasattr(Lh, "numel") else Lh.size) == n, "Lh has the wrong size"
        assert (sel.numel() if hasattr(sel, "numel") else sel.size) == ns * self.maxBatch
        self._cond_keep = None
        lib.nvw_set_inputs_n(self._h, addr(Lh), addr(sel), ns)

    # ---- beyond the reference class (include/nv_wavenet_c.h, "extensions") ---------------------
    def setConditioning(self, Lh, numSamples=None):
        """The conditioning half of setInputs; pair with setSelectorSeed (no selector matrix)."""
        Lh = _f32(Lh)
        ns = self.maxSamples if numSamples is None else int(numSamples)
        assert 0 < ns <= self.maxSamples
        n = ns * self.numLayers * self.maxBatch * 2 * self.R
        assert (Lh.numel() if hasattr(Lh, "numel") else Lh.size) == n, "Lh has the wrong size"
        self._cond_keep = None
        lib.nvw_set_conditioning_n(self._h, addr(Lh), ns)

    def setConditioningDirect(self, Lh, numSamples=None):
        """Device-resident conditioning consumed in place: Lh is a CUDA tensor [numSamples][L][maxBatch][2R], float32 or
        -- fp16 engines -- float16 (the engine's T_data: half the bytes); no packed copy is made and the kernels read it
        directly.  The tensor is kept referenced here until the next conditioning call; do not modify it while runs
        are in flight."""
        import torch
        assert hasattr(Lh, "data_ptr") and Lh.is_cuda and Lh.is_contiguous(), "setConditioningDirect takes contiguous device tensors"
        bits = {torch.float32: 32, torch.float16: 16}.get(Lh.dtype)
        if bits is None or (bits == 16 and self.precision != 16):
            raise TypeError("an fp%d engine cannot read a %s tensor in place" % (self.precision, Lh.dtype))
        ns = self.maxSamples if numSamples is None else int(numSamples)
        assert Lh.numel() == ns * self.numLayers * self.maxBatch * 2 * self.R, "Lh has the wrong size"
        self._cond_keep = Lh
        if not lib.nvw_set_conditioning_direct_t(self._h, addr(Lh), ns, bits):
            raise ValueError("nvw_set_conditioning_direct_t refused a %d-bit tensor of %d samples" % (bits, ns))

    def condTiles(self):
        """Tiles of 16 utterances per (sample, layer) row of the packed conditioning (the batch rounded up to whole workgroups)."""
        return int(lib.nvw_cond_tiles(self._h))

    def setConditioningPacked(self, frags, numSamples=None):
        """Conditioning already in the engine's fragment order: a contiguous CUDA tensor
        [numSamples + 1][L][condTiles()][waves][fragments][64][8 fp16 | 4 fp32] of the engine's T_data with the gate rows
        pre-scaled (nv_wavenet.py: pack_cond_input / get_cond_input(layout="packed") build it); used in place by the packed
        path of the generation kernels -- no copy, no conversion.  Kept referenced here until the next conditioning call."""
        import torch
        assert hasattr(frags, "data_ptr") and frags.is_cuda and frags.is_contiguous(), "setConditioningPacked takes contiguous device tensors"
        want = torch.float16 if self.precision == 16 else torch.float32
        if frags.dtype != want:
            raise TypeError("an fp%d engine reads %s fragments" % (self.precision, want))
        ns = self.maxSamples if numSamples is None else int(numSamples)
        assert 0 < ns <= self.maxSamples
        assert frags.numel() == (ns + 1) * self.numLayers * self.condTiles() * 16 * 2 * self.R, "fragment tensor has the wrong size"
        self._cond_keep = frags
        if not lib.nvw_set_conditioning_packed_n(self._h, addr(frags), ns, frags.numel()):
            raise ValueError("the fragment tensor is too small for %d samples" % ns)

    # ---- conditioning computed in the generation kernel from the upsampled features (include/nv_wavenet_c.h, round 5) ----------
    def setConditioningWeights(self, Wcond, bcond):
        """The model's `cond_layers` 1x1 convolution (pytorch/wavenet.py:73-74): Wcond [2R*L][n_cond] (or [2R*L][n_cond][1], or
        [L][2R][n_cond]), bcond [2R*L]; fp32, numpy or torch, host or device; copied."""
        W, b = _f32(Wcond), _f32(bcond)
        nW = W.numel() if hasattr(W, "numel") else W.size
        nb = b.numel() if hasattr(b, "numel") else b.size
        assert nb == self.numLayers * 2 * self.R and nW % nb == 0, "Wcond / bcond do not match L x 2R"
        self.nCond = nW // nb
        if not lib.nvw_set_conditioning_weights(self._h, addr(W), addr(b), self.nCond):
            raise ValueError("%d feature channels: the kernels are built for 1..%d" % (self.nCond, lib.nvw_max_cond_channels()))

    def featureFragments(self):
        return int(lib.nvw_feature_fragments(self._h))

    @staticmethod
    def _feat_args(x):
        import torch
        assert hasattr(x, "data_ptr") and x.is_cuda and x.dim() == 3, "features: a CUDA tensor [batch][n_cond][samples] (any strides)"
        bits = {torch.float32: 32, torch.float16: 16}.get(x.dtype)
        assert bits, "features must be float32 or float16"
        return bits, x.stride(0), x.stride(1), x.stride(2)

    def setFeatures(self, x, numSamples=None):
        """The upsampled features of the whole utterance: CUDA tensor [maxBatch][n_cond][numSamples] (the model's upsample
        output; any strides, e.g. a channels-last tensor transposed).  Copied into fragment order; resets the history."""
        bits, sb, sc, st = self._feat_args(x)
        ns = x.size(2) if numSamples is None else int(numSamples)
        assert x.size(0) == self.maxBatch and x.size(1) == self.nCond and x.size(2) >= ns
        self._cond_keep = None
        if not lib.nvw_set_features(self._h, x.data_ptr(), bits, sb, sc, st, ns):
            raise ValueError("nvw_set_features refused %s (%d samples)" % (tuple(x.shape), ns))

    def packFeatures(self, x, firstSample, stream=None):
        """Samples [firstSample, firstSample + x.size(2)) of the features, asynchronously on `stream`; history untouched."""
        bits, sb, sc, st = self._feat_args(x)
        assert x.size(0) == self.maxBatch and x.size(1) == self.nCond
        self._cond_keep = None
        if not lib.nvw_pack_features(self._h, x.data_ptr(), bits, sb, sc, st, int(firstSample), x.size(2), stream):
            raise ValueError("nvw_pack_features refused samples [%d, %d)" % (firstSample, firstSample + x.size(2)))

    def setConditioningFeatures(self, frags, numSamples=None):
        """Features already in fragment order: CUDA tensor [numSamples][condTiles()][featureFragments()][64][8 fp16 | 4 fp32] of the
        engine's T_data (nv_wavenet.py:feature_fragments builds it); used in place."""
        import torch
        assert hasattr(frags, "data_ptr") and frags.is_cuda and frags.is_contiguous()
        want = torch.float16 if self.precision == 16 else torch.float32
        assert frags.dtype == want, "an fp%d engine reads %s fragments" % (self.precision, want)
        ns = frags.size(0) if numSamples is None else int(numSamples)
        self._cond_keep = frags
        if not lib.nvw_set_conditioning_features(self._h, addr(frags), ns, frags.numel()):
            raise ValueError("the feature fragment tensor does not fit %d samples" % ns)

    # ---- features in, samples out: the upsampling on the engine's own kernel and the streaming loop ---------------------------
    def setUpsampling(self, upW, upB, stride):
        """The model's `upsample` ConvTranspose1d (pytorch/wavenet.py:70-72): weight [n_cond][n_cond][window], bias [n_cond]."""
        W, b = _f32(upW), _f32(upB)
        window = int(W.shape[2])
        assert tuple(W.shape[:2]) == (self.nCond, self.nCond), "upsample weight must be [n_cond][n_cond][window]"
        if not lib.nvw_set_upsampling(self._h, addr(W), addr(b), window, int(stride)):
            raise ValueError("upsampling window %d / stride %d not supported" % (window, stride))
        self.upStride = int(stride)

    def setMel(self, mel):
        """The utterances' frames before upsampling: CUDA tensor [maxBatch][n_cond][frames], float32 or float16, any strides.  Copied;
        resets the history (the start of an utterance batch)."""
        bits, sb, sc, sf = self._feat_args(mel)
        assert mel.size(0) == self.maxBatch and mel.size(1) == self.nCond
        self._cond_keep = None
        if not lib.nvw_set_mel(self._h, mel.data_ptr(), bits, sb, sc, sf, mel.size(2)):
            raise ValueError("nvw_set_mel refused %s" % (tuple(mel.shape),))
        self.melFrames = mel.size(2)

    def upsampleFeatures(self, firstSample, count, stream=None):
        if not lib.nvw_upsample_features(self._h, int(firstSample), int(count), stream):
            raise ValueError("nvw_upsample_features refused samples [%d, %d)" % (firstSample, firstSample + count))

    def getFeatures(self, firstSample, count):
        """Debug getter: the engine's own feature fragments of samples [firstSample, firstSample + count) as a CUDA tensor
        [count][condTiles()][featureFragments()][4][16][8 | 4]."""
        import torch
        epl = 8 if self.precision == 16 else 4
        out = torch.empty(count, self.condTiles(), self.featureFragments(), 4, 16, epl, device="cuda",
                          dtype=torch.float16 if self.precision == 16 else torch.float32)
        if not lib.nvw_get_features(self._h, out.data_ptr(), int(firstSample), int(count)):
            raise ValueError("nvw_get_features refused samples [%d, %d)" % (firstSample, firstSample + count))
        return out

    def generate_stream(self, num_samples_per_chunk, consume, num_samples, batch_size, yOut=None, stream=None):
        """Features in, samples out: per chunk the upsampling, the generation launch and the copy of the chunk's samples (and PCM);
        consume(yOut, first, count) on this thread for every finished chunk."""
        def _cb(_ptr, init, count, _user):
            if consume is not None:
                consume(yOut, init, count)
        cb = CONSUME_FN(_cb)
        self._cb_keep = cb
        return bool(lib.nvw_generate_stream(self._h, int(num_samples_per_chunk), cb, None, int(num_samples), int(batch_size),
                                            self._yout(yOut, num_samples), stream))

    def setSelectors(self, outputSelectors, numSamples=None):
        """The selector half of setInputs: [numSamples][maxBatch] uniform draws; conditioning and history untouched."""
        sel = _f32(outputSelectors)
        ns = self.maxSamples if numSamples is None else int(numSamples)
        assert (sel.numel() if hasattr(sel, "numel") else sel.size) == ns * self.maxBatch
        lib.nvw_set_selectors(self._h, addr(sel), ns)

    def packConditioning(self, Lh, firstSample, count, stream=None):
        """Conditioning streamed chunk by chunk: Lh [count][L][maxBatch][2R] fp32 ON THE DEVICE holds samples
        firstSample .. firstSample+count-1; packed asynchronously on `stream` (history untouched)."""
        Lh = _f32(Lh)
        assert hasattr(Lh, "data_ptr") and Lh.is_cuda, "packConditioning takes device tensors"
        assert Lh.numel() == count * self.numLayers * self.maxBatch * 2 * self.R
        self._cond_keep = None
        lib.nvw_pack_conditioning(self._h, addr(Lh), int(firstSample), int(count), stream)

    def run_partial_chunk(self, init_sample, count, num_samples, batch_size, stream=None):
        """Samples [init_sample, init_sample+count) of a num_samples-long utterance, asynchronously on `stream`."""
        return bool(lib.nvw_run_range(self._h, int(init_sample), int(count), int(num_samples), int(batch_size), stream))

    def resetHistory(self, stream=None):
        lib.nvw_reset_history(self._h, stream)

    def chainStatus(self):
        """0, or the code of a multi-CU launch that gave up and could not be re-run (synchronises)."""
        return int(lib.nvw_chain_status(self._h))

    def chainFallbacks(self):
        """Multi-CU launches that gave up (not all workgroups resident in time) and were re-run on wavenet_wg."""
        return int(lib.nvw_chain_fallbacks(self._h))

    def chainLastTimeout(self):
        return int(lib.nvw_chain_last_timeout(self._h))

    def setChainTimeoutMs(self, ms):
        lib.nvw_set_chain_timeout_ms(self._h, float(ms))

    def setRingInLds(self, mode=0):
        """The dilation ring on chip: mode >= 0 (default) = single-workgroup launches keep the ring slots of as many short-dilation layers
        as fit in LDS; -1 = never.  Same samples either way."""
        lib.nvw_set_ring_in_lds(self._h, int(mode))

    def setClockProbe(self, on=True):
        """Measurement aid: workgroup 0 of every wavenet_wg launch that follows records shader and wall clock counters."""
        lib.nvw_set_clock_probe(self._h, 1 if on else 0)

    def lastLaunchClockGHz(self):
        """Shader clock the latest probed launch actually ran at (GHz; 0.0 if none was probed); synchronises."""
        return float(lib.nvw_last_launch_clock_ghz(self._h))

    def kernelInfo(self, batch_size=None, dumpActivations=False):
        """The device code run(n, batch_size, ..., dumpActivations) launches (kernel + template arguments)."""
        import ctypes
        buf = ctypes.create_string_buffer(256)
        lib.nvw_kernel_info(self._h, self.maxBatch if batch_size is None else batch_size, 1 if dumpActivations else 0,
                            buf, 256)
        return buf.value.decode()

    def setSelectorSeed(self, seed):
        """Selectors are drawn in-kernel (Philox4x32-10, counter {sample, utterance, 0, 0}, key=seed)."""
        lib.nvw_set_selector_seed(self._h, int(seed) & 0xFFFFFFFFFFFFFFFF)

    def setAudioOut(self, pcmOut):
        """pcmOut: int16 [batch][num_samples] of the run calls that follow (numpy or CUDA tensor) filled wherever yOut is with
        int16(32768 * mu_law_decode(y, A)); None switches it off."""
        if pcmOut is not None:
            if hasattr(pcmOut, "data_ptr"):
                import torch
                assert pcmOut.dtype == torch.int16
            else:
                assert pcmOut.dtype == np.int16 and pcmOut.flags["C_CONTIGUOUS"]
            n = pcmOut.numel() if hasattr(pcmOut, "numel") else pcmOut.size
            self._pcm_keep = pcmOut
            lib.nvw_set_audio_out_n(self._h, addr(pcmOut), n)   # every run call checks batch * num_samples <= n
        else:
            self._pcm_keep = None
            lib.nvw_set_audio_out(self._h, None)

    # ---- run --------------------------------------------------------------------------------
    def _yout(self, yOut, need=None):
        if yOut is None:
            return None
        if hasattr(yOut, "data_ptr"):
            import torch
            assert yOut.dtype == torch.int32
        else:
            assert yOut.dtype == np.int32 and yOut.flags["C_CONTIGUOUS"]
        n = yOut.numel() if hasattr(yOut, "numel") else yOut.size
        assert n >= self.maxBatch * (self.maxSamples if need is None else need), "yOut must hold [maxBatch][num_samples] int32"
        return addr(yOut)

    def run(self, num_samples, batch_size, yOut=None, batch_size_per_block=1, dumpActivations=False, stream=None):
        ok = lib.nvw_run(self._h, num_samples, batch_size, self._yout(yOut, num_samples), batch_size_per_block,
                         1 if dumpActivations else 0, stream)
        return bool(ok)

    def run_partial(self, init_sample, num_samples, batch_size, yOut=None, batch_size_per_block=1,
                    dumpActivations=False, stream=None):
        ok = lib.nvw_run_partial(self._h, init_sample, num_samples, batch_size, self._yout(yOut, num_samples),
                                 batch_size_per_block, 1 if dumpActivations else 0, stream)
        return bool(ok)

    def run_chunks(self, num_samples_per_chunk, consume, num_samples, batch_size, yOut=None,
                   batch_size_per_block=1, dumpActivations=False, stream=None):
        """consume(yOut, init_sample, count) is called on this thread for every finished chunk."""
        def _cb(_ptr, init, count, _user):
            if consume is not None:
                consume(yOut, init, count)
        cb = CONSUME_FN(_cb)
        self._cb_keep = cb
        ok = lib.nvw_run_chunks(self._h, num_samples_per_chunk, cb, None, num_samples, batch_size,
                                self._yout(yOut, num_samples), batch_size_per_block, 1 if dumpActivations else 0, stream)
        return bool(ok)

    # ---- getters (host numpy, reference layouts) ----------------------------------------------
    def _get(self, fn, shape, *pre):
        out = np.zeros(shape, dtype=np.float32)
        fn(self._h, *pre, addr(out))
        return out

    def getXtOut(self, layer):
        return self._get(lib.nvw_get_xt_out, (self.maxBatch, self.R), layer)

    def getSkipOut(self, layer):
        return self._get(lib.nvw_get_skip_out, (self.maxBatch, self.S), layer)

    def getZs(self):
        return self._get(lib.nvw_get_zs, (self.maxBatch, self.A))

    def getZa(self):
        return self._get(lib.nvw_get_za, (self.maxBatch, self.A))

    def getP(self):
        return self._get(lib.nvw_get_p, (self.maxBatch, self.A))

    def getYOut(self, yOut, offset, size, stream=None):
        lib.nvw_get_y_out(self._h, self._yout(yOut), offset, size, stream)

    def time_runs(self, reps, num_samples, batch_size, batch_size_per_block=1, stream=None):
        """HIP-event milliseconds for `reps` back-to-back run() launches on `stream`."""
        return float(lib.nvw_time_runs(self._h, reps, num_samples, batch_size, batch_size_per_block, stream))

    @staticmethod
    def synchronize():
        lib.nvw_device_synchronize()
