#!/usr/bin/env python3
"""Summary of the per-wavefront milestones a -DO3D_PASS_PROF=1 build of icp.hip dumps (O3D_PASS_PROF_DUMP=file, one record of
4096 x 8 uint64 per pass; scripts/pass_phases.sh): when the wavefronts start and end their certificate loop, by CU, SIMD and
by the order in which their workgroup was dispatched on its XCD.  usage: pass_marks_summary.py gpurun_out/pass_marks.bin"""
import sys
import numpy as np

a = np.fromfile(sys.argv[1], dtype=np.uint64)
marks = 24 if a.size % (4096 * 24) == 0 else (16 if a.size % (4096 * 16) == 0 else 8)
a = a.reshape(-1, 4096, marks)
for ps in range(a.shape[0]):
    m = a[ps]
    t0 = m[:, 0].min()
    t = (m[:, :6].astype(np.int64) - int(t0)) / 100.0  # microseconds after the first wavefront came in
    hw = m[:, 6] & 0xFFFFFFFF
    xcc = (m[:, 6] >> 32) & 0xF
    simd, cu, sh, se = (hw >> 4) & 3, (hw >> 8) & 0xF, (hw >> 12) & 1, (hw >> 13) & 7
    if t[:, 2].max() <= 0:
        print("pass %d: the certificate half stood aside; all done mean %.1f latest %.1f us" % (ps, t[:, 5].mean(), t[:, 5].max()))
        continue
    loop = t[:, 2]
    cuid = xcc * 1000 + se * 100 + sh * 16 + cu
    ids = np.unique(cuid)
    cm = np.array([loop[cuid == c].mean() for c in ids])
    cnt = np.array([(cuid == c).sum() for c in ids])
    wg = (np.arange(4096) // 4) >> 3  # workgroup's number on its XCD (dispatch order)
    print("pass %d: loop starts mean %.1f latest %.1f | loop done mean %.1f earliest %.1f latest %.1f | all done latest %.1f us"
          % (ps, t[:, 1].mean(), t[:, 1].max(), loop.mean(), loop.min(), loop.max(), t[:, 5].max()))
    print("   %d CUs with %d..%d wavefronts each; per-CU mean of loop done %.1f..%.1f; by SIMD %s; by XCD %s"
          % (len(ids), cnt.min(), cnt.max(), cm.min(), cm.max(), [round(float(loop[simd == k].mean()), 1) for k in range(4)],
             [round(float(loop[xcc == k].mean()), 1) for k in np.unique(xcc)]))
    print("   loop done by dispatch order of the workgroup on its XCD (quarters: the 1st, 2nd, 3rd, 4th workgroup of a CU): %s"
          % [round(float(loop[(wg // 32) == k].mean()), 1) for k in range(4)])
    if marks >= 16:
        ph = m[:, 8:13].astype(np.float64)
        tot = ph.sum(1)
        names = ["top: wait for positions, issue loads", "update + write-back (waits for the point)", "wait for the record, distance", "certificate test, bound, work list", "sums"]
        print("   shader-clock cycles per wavefront and trip (mean over wavefronts; 38.15 trips): total %.0f" % (tot.mean() / 38.15))
        for i, nm in enumerate(names):
            print("      %-46s %7.0f  (%4.1f %%)" % (nm, ph[:, i].mean() / 38.15, 100 * ph[:, i].mean() / tot.mean()))
        last = np.nonzero(m[:, 7])[0]
        if len(last):
            w = last[0]
            t5 = lambda i: (int(m[w, i]) - int(t0)) / 100.0
            print("   the workgroup that ends the pass (us after the first wavefront came in): own search done and group rows added %.1f, all rows added and state read %.1f, "
                  "estimate solved %.1f, out %.1f" % (t5(14), t5(15), t5(7), t5(5)))
            if marks >= 24 and m[w, 20]:
                print("      the estimate on its one lane: criteria %.1f, 6 x 6 LDLT %.1f, angles to matrix %.1f, composition %.1f, certificate forecast %.1f us"
                      % (t5(16) - t5(15), t5(17) - t5(16), t5(18) - t5(17), t5(19) - t5(18), t5(20) - t5(19)))
