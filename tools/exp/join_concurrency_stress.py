#!/usr/bin/env python3
"""Does the persistent fused join (k_row_join_fused) survive OTHER work on the GPU at the same time?  Its grid is what the
device holds when it is alone; a second stream's kernels take wave slots away while it runs.  Counts look-back time-outs
(gpc_hip_synchronize -> GPC_E_HIP) per launches and checks the outputs of the launches that did not report one.
modes: alone | torch (an unrelated elementwise kernel loop on a second stream) | wired (the next batch's k_preprocess beside
the join: tools/overlap_experiment.py's wiring).   usage: join_concurrency_stress.py MODE [launches] [pairs]"""
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

import opengpc_amd as g  # noqa: E402
from opengpc_amd.synth import synth_batch  # noqa: E402


def main():
    mode = sys.argv[1] if len(sys.argv) > 1 else "alone"
    launches = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
    B = int(sys.argv[3]) if len(sys.argv) > 3 else 256
    W, H = 1024, 436
    dev = torch.device("cuda", 0)
    L, R = synth_batch(W, H, list(range(B)))
    d_L, d_R = torch.from_numpy(L).to(dev), torch.from_numpy(R).to(dev)
    cap = (W - 26) * (H - 26)
    s = g.Settings.sparsematch()
    lanes = []
    for i in range(2 if mode in ("wired", "lanes") else 1):
        c = g.Context(0)
        st = torch.cuda.Stream(device=dev)
        c.set_stream(st.cuda_stream)
        c.load_forest(os.path.join(ROOT, "forests", "defaultZeroForest.txt"), W, H)
        c.reserve(W, H, B)
        out = torch.zeros((B, cap, 3), dtype=torch.int32, device=dev)
        cnt = torch.zeros(B, dtype=torch.int32, device=dev)
        nc = torch.zeros((B, 2), dtype=torch.int32, device=dev)
        eh, ej = torch.cuda.Event(), torch.cuda.Event()
        eh.record(st)
        ej.record(st)
        lanes.append([c, st, out, cnt, nc, eh, ej])
        if mode == "lanes" and i == 1:
            # ONE context with the library's two lanes: both "lanes" of this script are output sets of context 0
            c.close()
            lanes[1][0] = lanes[0][0]
            lanes[0][0].set_pipeline(2)
    torch.cuda.synchronize(dev)
    # reference outputs: one launch alone
    c0 = lanes[0]
    c0[0].match_batch_device(d_L.data_ptr(), d_R.data_ptr(), W, H, B, s, c0[2].data_ptr(), cap, c0[3].data_ptr(), c0[4].data_ptr())
    c0[0].synchronize()
    ref_out, ref_cnt = c0[2].clone(), c0[3].clone()
    if mode == "wired":
        fn = c0[0].L.gpc_hip_debug_pipeline_events
        fn.argtypes = [C.c_void_p] * 5
        for i, ln in enumerate(lanes):
            o = lanes[1 - i]
            fn(ln[0].h, C.c_void_p(o[5].cuda_event), C.c_void_p(o[6].cuda_event), C.c_void_p(ln[5].cuda_event), C.c_void_p(ln[6].cuda_event))
    side = torch.cuda.Stream(device=dev)
    junk = torch.ones(64 << 20, dtype=torch.float32, device=dev)
    errors, bad, done = 0, 0, 0
    t0 = time.time()
    k = 0
    while done < launches:
        chunk = int(os.environ.get('STRESS_CHUNK', '20'))
        for _ in range(chunk):
            ln = lanes[k % len(lanes)]
            k += 1
            if mode == "torch":
                with torch.cuda.stream(side):
                    junk.mul_(1.0001)
            ln[0].match_batch_device(d_L.data_ptr(), d_R.data_ptr(), W, H, B, s, ln[2].data_ptr(), cap, ln[3].data_ptr(), ln[4].data_ptr())
        done += chunk
        failed = False
        for ln in lanes:
            try:
                ln[0].synchronize()
            except g.capi.GpcError as e:
                errors += 1
                failed = True
                print("after %d launches: %s" % (done, str(e)[:170]), flush=True)
        torch.cuda.synchronize(dev)
        if not failed:
            for ln in lanes:
                if not (torch.equal(ln[3], ref_cnt) and torch.equal(ln[2], ref_out)):
                    bad += 1
                    print("after %d launches: outputs differ WITHOUT a reported time-out" % done, flush=True)
        if time.time() - t0 > 240:
            break
    print("mode %s, %d pairs: %d launches, %d reported look-back time-outs, %d silent differences, %.1f s" % (mode, B, done, errors, bad, time.time() - t0))
    for ln in lanes:
        try:
            ln[0].close()
        except Exception:
            pass


if __name__ == "__main__":
    main()
