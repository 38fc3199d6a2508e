"""Python module with the surface of the reference's pybind extension ``nv_wavenet_ext``
(/root/reference/pytorch/wavenet_infer_wrapper.cpp:32-110), implemented over the C ABI
``wavenet_infer`` / ``get_R`` / ``get_S`` / ``get_A`` of libwavenet_infer.so instead of pybind:
same function names, argument order and meaning; tensors are torch tensors (host or device) or
numpy arrays, fp32, already column-major as pytorch/nv_wavenet.py:33-49 prepares them.
"""
import ctypes as C

from ._lib import lib, addr


def _check(t, name, dtype):
    have = str(t.dtype).replace("torch.", "")
    if have != dtype:
        raise TypeError("%s must be %s, got %s" % (name, dtype, have))
    contiguous = t.is_contiguous() if hasattr(t, "is_contiguous") else t.flags["C_CONTIGUOUS"]
    if not contiguous:
        raise ValueError("%s must be contiguous" % name)


def infer(samples_tensor, sample_count, batch_size, embed_prev_tensor, embed_curr_tensor, conv_out_tensor,
          conv_end_tensor, cond_input_tensor, num_layers, use_embed_tanh, max_dilation, implementation, layers):
    """layers: flat list, 7 tensors per layer: Wprev, Wcur, Bh, Wres, Bres, Wskip, Bskip
    (wavenet_infer_wrapper.cpp:60-69). Fills samples_tensor [batch][samples] int32; returns 1."""
    assert len(layers) == 7 * num_layers, "expected 7 tensors per layer"
    # the reference's pybind wrapper reads tensor.data<float>() / data<int>(), which throws on any other dtype;
    # raw addresses would silently reinterpret a half / double model or an int64 sample buffer
    _check(samples_tensor, "samples_tensor", "int32")
    for name, t in (("embed_prev_tensor", embed_prev_tensor), ("embed_curr_tensor", embed_curr_tensor),
                    ("conv_out_tensor", conv_out_tensor), ("conv_end_tensor", conv_end_tensor),
                    ("cond_input_tensor", cond_input_tensor)):
        _check(t, name, "float32")
    for i, t in enumerate(layers):
        _check(t, "layers[%d]" % i, "float32")
    ptrs = []
    for k in range(7):
        arr = (C.c_void_p * num_layers)()
        for i in range(num_layers):
            arr[i] = addr(layers[i * 7 + k])
        ptrs.append(arr)
    lib.wavenet_infer(sample_count, batch_size, addr(embed_prev_tensor), addr(embed_curr_tensor), num_layers,
                      max_dilation, ptrs[0], ptrs[1], ptrs[2], ptrs[3], ptrs[4], ptrs[5], ptrs[6],
                      addr(conv_out_tensor), addr(conv_end_tensor), int(use_embed_tanh), addr(cond_input_tensor),
                      int(implementation), addr(samples_tensor))
    return 1


def num_res_channels():
    return lib.get_R()


def num_skip_channels():
    return lib.get_S()


def num_out_channels():
    return lib.get_A()
