"""Round 5: where the time of the features-in loop goes (C3 fp16): upsampling alone, generation (RAW=3) alone, the streamed loop with and
without the per-chunk sample copies.  usage: gpu_r5_stream.py [batch]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import torch  # noqa: E402


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 12288
    chunk, chunks = 256, 4
    N = chunk * chunks
    w = bench.make_weights()
    e = bench.build_engine(w, B, N)
    Wc, bc = bench.make_cond_layers()
    e.setConditioningWeights(Wc, bc)
    rng = np.random.default_rng(1)
    up_w = ((rng.random((80, 80, bench.UP_WINDOW), dtype=np.float32) - 0.5) * 0.2).astype(np.float32)
    e.setUpsampling(up_w, np.zeros(80, dtype=np.float32), bench.UP_STRIDE)
    mel = torch.randn(B, 80, N // bench.UP_STRIDE, device="cuda").half()
    e.setSelectorSeed(3)
    e.setMel(mel)
    st = torch.cuda.current_stream()

    def timed(fn, reps=3):
        best = 1e9
        for _ in range(reps):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            fn()
            torch.cuda.synchronize()
            best = min(best, 1e3 * (time.perf_counter() - t0))
        return best
    e.upsampleFeatures(0, N)
    e.run_partial_chunk(0, 64, N, B, st.cuda_stream)
    torch.cuda.synchronize()
    print("upsampling of one chunk (%d samples x %d utterances): %.3f ms" % (chunk, B, timed(lambda: e.upsampleFeatures(chunk, chunk, st.cuda_stream))))
    print("upsampling of four chunks at once: %.3f ms" % timed(lambda: e.upsampleFeatures(0, N, st.cuda_stream)))

    def gen_all():
        e.resetHistory(st.cuda_stream)
        for j in range(chunks):
            e.run_partial_chunk(j * chunk, chunk, N, B, st.cuda_stream)
    t_gen = timed(gen_all)
    print("generation only, %d chunks: %.3f ms = %.2f kHz" % (chunks, t_gen, N / t_gen))

    def both():
        e.resetHistory(st.cuda_stream)
        for j in range(chunks):
            e.upsampleFeatures(j * chunk, chunk, st.cuda_stream)
            e.run_partial_chunk(j * chunk, chunk, N, B, st.cuda_stream)
    t_b = timed(both)
    print("upsampling + generation per chunk, one stream: %.3f ms = %.2f kHz" % (t_b, N / t_b))
    y = torch.zeros(B, N, dtype=torch.int32, device="cuda")
    for label, yy in (("no sample copies", None), ("samples copied per chunk", y)):
        def stream():
            e.setMel(mel)
            assert e.generate_stream(chunk, None, N, B, yy)
        t_s = timed(stream)
        print("nvw_generate_stream (incl. nvw_set_mel), %s: %.3f ms = %.2f kHz" % (label, t_s, N / t_s))
    print("nvw_set_mel alone: %.3f ms" % timed(lambda: e.setMel(mel)))
    e.close()


if __name__ == "__main__":
    main()
