"""diagnostic (round 6): the random MFCC configuration whose fast-mode feature error exceeded tests/test_gpu_fast_mode.py's bound after the guard's re-fit"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from __graft_entry__ import load_package
from kws_testlib import L476_CONFIG, Oracle, random_dsp_spec, synth_model_blob
from kws_families import column_conditioning
pkg = load_package()
oracle = Oracle()
clips = oracle.synth(21, 0, 24)
pcm = torch.from_numpy(clips).to("cuda:0")
for seed in [int(a) for a in sys.argv[1:]] or [15]:
    cfg_kw, blob_kw = random_dsp_spec(seed)
    gm = pkg.Model(blob=synth_model_blob(**blob_kw))
    gm.set_mode(pkg.MODE_FAST)
    cfg = L476_CONFIG().copy(**cfg_kw)
    n = len(clips)
    s = torch.zeros((n, gm.n_labels), dtype=torch.float32, device="cuda:0")
    f = torch.zeros((n, gm.n_features), dtype=torch.float32, device="cuda:0")
    q = torch.zeros((n, gm.n_features), dtype=torch.int8, device="cuda:0")
    gm.run_classifier_batch_device(pcm.data_ptr(), n, s.data_ptr(), f.data_ptr(), q.data_ptr())
    torch.cuda.synchronize()
    f = f.cpu().numpy()
    fo = np.stack([oracle.extract_mfcc(c, cfg) for c in clips])
    cep = np.stack([oracle.mfcc_nocmvn(c, cfg) for c in clips])
    sdw, mw = column_conditioning(cep, cfg.win_size, full=True)
    nfr, nc = sdw.shape[1], sdw.shape[2]
    d = np.abs(f - fo).reshape(n, nfr, nc)
    i, r, c = np.unravel_index(np.argmax(d), d.shape)
    coef = gm.fast_guard(1)
    tol = gm.fast_tolerance()
    print("seed", seed, cfg_kw, "fallback", gm.fast_fallback_count(), "tolerance", {k: tol[k] for k in ("calibrated", "silent_rows_exact", "sigma_net", "total_gain")})
    print("  worst |df| %.3g at clip %d row %d col %d: window sd %.3g mean %.3g; cepstral-domain error %.3g; coef abs/lev/rel/alt of that column: %s" % (d[i, r, c], i, r, c, sdw[i, r, c], mw[i, r, c], d[i, r, c] * sdw[i, r, c], coef[:, c]))
    mel = np.stack([np.abs(np.log(oracle.mfe(x, cfg)[0].astype(np.float64)).mean(axis=1)).mean() for x in clips])
    print("  level of that clip %.2f; per-column rms over the batch of |df| x sd / level (x1e7): %s" % (mel[i], np.array2string(np.sqrt(((d * sdw) ** 2).mean(axis=(0, 1))) / mel.mean() * 1e7, precision=2, max_line_width=200)))
    gm.close()
