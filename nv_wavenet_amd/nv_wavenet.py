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
        self.R = nv_wavenet_ext.num_res_channels()
        self.S = nv_wavenet_ext.num_skip_channels()
        self.A = nv_wavenet_ext.num_out_channels()
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
