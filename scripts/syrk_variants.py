"""k_syrk geometry sweep (development aid): MCBA_SYRK_G x MCBA_SYRK_FS."""
import os
import sys

sys.path.insert(0, ".")
import multicam_calibration_amd as m

C, F = int(sys.argv[1]), int(sys.argv[2])
rows, cols = (int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else (6, 9)
p = m.synth.make_problem(C, F, rows=rows, cols=cols, seed=0)
x0 = m.serialize_params(p["extrinsics"], p["intrinsics"], p["poses"])
for G in (256, 512, 768, 1024):
    for FS in (16, 8, 4):
        os.environ["MCBA_SYRK_G"], os.environ["MCBA_SYRK_FS"] = str(G), str(FS)
        prob = m.ops.Problem(p["uvs"], p["obj"])
        prob.set_params(0, x0)
        prob.linearize(0)
        prob.profile_enable(True)
        for it in range(12):
            prob.build_reduced(1e-3)
            if it == 1:
                prob.profile_read()
        prob.synchronize()
        pr = prob.profile_read()
        print("C=%d F=%d G<=%d FS=%d: k_syrk %.1f us  k_reduce %.1f us" % (C, F, G, FS, 1e3 * pr["k_syrk"][0] / pr["k_syrk"][1], 1e3 * pr["k_reduce_system"][0] / pr["k_reduce_system"][1]), flush=True)
        prob.close()
