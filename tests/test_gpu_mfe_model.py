"""-m gpu: a model whose DSP block is MFE (extract_mfe_features / extract_mfe_per_slice_features /
calc_cepstral_mean_and_var_normalization_mfe of the newer SDK copy) through every entry point of the library, against the golden
vectors composed from the reference's own leaves (tests/golden/mfe_model_l432.npz, tools/make_golden.py) and the oracle."""
import os
import subprocess
import sys

import numpy as np
import pytest

from kws_testlib import GOLDEN, ROOT, OracleContinuous, OracleModel, bits, special_clips, synth_model_blob

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def pkg():
    sys.path.insert(0, ROOT)
    import torch  # noqa: F401
    from __graft_entry__ import load_package
    return load_package()


def mfe_kw():
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from make_golden import MFE_MODEL_KW
    return MFE_MODEL_KW


def test_mfe_model_batch_golden_and_oracle(pkg, oracle, tmp_path):
    g = np.load(os.path.join(GOLDEN, "mfe_model_l432.npz"))
    blob = synth_model_blob(**mfe_kw())
    gm = pkg.Model(blob=blob)
    assert gm.n_features == 49 * 32
    sp = special_clips()
    clips = np.concatenate([oracle.synth(int(g["seed"]), int(g["first"]), int(g["n"])), np.stack([sp[str(k)] for k in g["special_names"]])])
    s, f, q = gm.run_classifier_batch(clips, want_features=True)
    assert (bits(f) == bits(g["features"])).all() and (q == g["q"]).all() and (bits(s) == bits(g["scores"])).all()
    assert (bits(gm.run_classifier_batch(clips)) == bits(g["scores"])).all()          # scores only (host path keeps its own float buffer)
    # more clips against the oracle, int8 and float32 twin, 40 filters too
    path = str(tmp_path / "m.kwsm")
    for kw in (mfe_kw(), dict(mfe_kw(), num_filters=40, high=0, win_size=51, seed=78)):
        open(path, "wb").write(synth_model_blob(**kw))
        om, gm2 = OracleModel(oracle, path), pkg.Model(path)
        c2 = oracle.synth(41, 0, 300)
        s2, f2, q2 = gm2.run_classifier_batch(c2, want_features=True)
        so, fo, qo = om.run_batch(c2, want_features=True)
        assert (bits(f2) == bits(fo)).all() and (q2 == qo).all() and (bits(s2) == bits(so)).all()
        pf = str(tmp_path / "f32.kwsm")
        subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "dequantize_model.py"), path, pf])
        omf, gmf = OracleModel(oracle, pf), pkg.Model(pf)
        sf, ff, _ = gmf.run_classifier_batch(c2[:64], want_features=True)
        sfo, ffo, _ = omf.run_batch(c2[:64], want_features=True)
        assert (bits(ff) == bits(ffo)).all() and np.abs(sf - sfo).max() <= 1e-6
        # KWS_MODE_FAST for the MFE block (round 3), device entry points (the host-buffer calls stay exact): the front end in tolerance
        # arithmetic (kws_fast_kernel<..., MFE>), then the block's own normalisation and the network on the exact kernels.  Nothing is
        # divided by a deviation: no clip is handed back, and the only degenerate input -- silence: every mel energy FLT_EPSILON, range
        # 0 -- comes out as the reference's own 0 x inf.
        import torch
        c3 = np.concatenate([oracle.synth(43, 0, 1024), np.stack([special_clips()[k] for k in ("zeros", "impulses", "alternating_fullscale")])])
        d3 = torch.from_numpy(c3).to("cuda:0")

        def run_dev(m, mode, want_f=True):
            m.set_mode(mode)
            n = len(c3)
            sc = torch.zeros((n, m.n_labels), dtype=torch.float32, device="cuda:0")
            ft = torch.zeros((n, m.n_features), dtype=torch.float32, device="cuda:0")
            qt = None if m.is_float else torch.zeros((n, m.n_features), dtype=torch.int8, device="cuda:0")
            m.run_classifier_batch_device(d3.data_ptr(), n, sc.data_ptr(), ft.data_ptr() if want_f else None, qt.data_ptr() if (qt is not None and want_f) else None)
            torch.cuda.synchronize()
            return sc.cpu().numpy(), ft.cpu().numpy(), (qt.cpu().numpy() if qt is not None else None)
        sfo3, ffo3, _ = omf.run_batch(c3, want_features=True)
        ok = np.isfinite(ffo3).all(axis=1)
        assert ok.sum() >= 1024
        sff, fff, _ = run_dev(gmf, pkg.MODE_FAST)
        assert (np.isfinite(fff).all(axis=1) == ok).all()
        assert (bits(fff[ok]) != bits(ffo3[ok])).any()                                           # it IS the other arithmetic
        assert np.abs(fff[ok] - ffo3[ok]).max() <= 2e-5 and np.abs(sff[ok] - sfo3[ok]).max() <= 1e-4, (np.abs(fff[ok] - ffo3[ok]).max(), np.abs(sff[ok] - sfo3[ok]).max())
        assert gmf.fast_fallback_count() == 0
        assert (bits(run_dev(gmf, pkg.MODE_FAST, want_f=False)[0][ok]) == bits(sff[ok])).all()   # scores only: the same launches
        se3, fe3, _ = run_dev(gmf, pkg.MODE_EXACT)
        assert (bits(fe3[ok]) == bits(ffo3[ok])).all() and np.abs(se3[ok] - sfo3[ok]).max() <= 1e-6
        so3, fo3, qo3 = om.run_batch(c3, want_features=True)
        s3, f3, q3 = run_dev(gm2, pkg.MODE_FAST)
        assert np.abs(f3[ok] - fo3[ok]).max() <= 2e-5
        assert np.abs(q3[ok].astype(np.int32) - qo3[ok].astype(np.int32)).max() <= 1 and (q3[ok] != qo3[ok]).mean() <= 1e-3
        for k in np.nonzero(ok)[0][:64]:                                                         # exact from the int8 tensor on
            assert (bits(om.dequantize(om.nn_invoke(q3[k]))) == bits(s3[k])).all(), k
        s4, f4, q4 = run_dev(gm2, pkg.MODE_EXACT)                                                # and back
        assert (bits(f4[ok]) == bits(fo3[ok])).all() and (q4[ok] == qo3[ok]).all() and (bits(s4[ok]) == bits(so3[ok])).all()
        gm2.close(); gmf.close()
    gm.close()


def test_mfe_model_streams_follow_the_oracle(pkg, oracle, tmp_path):
    import torch
    path = str(tmp_path / "m.kwsm")
    open(path, "wb").write(synth_model_blob(**mfe_kw()))
    om, gm = OracleModel(oracle, path), pkg.Model(path)
    S, n_steps = 21, 10
    audio = oracle.synth(15, 0, S * 3).reshape(S, 3 * 16000)
    sb = pkg.StreamBatch(gm, S)
    ocs = [OracleContinuous(om) for _ in range(S)]
    for oc in ocs:
        oc.init()
    scores = torch.empty((S, gm.n_labels), dtype=torch.float32, device="cuda")
    n_produced = 0
    for k in range(n_steps):
        sl = np.ascontiguousarray(audio[:, k * 4000:(k + 1) * 4000])
        d = torch.from_numpy(sl).cuda()
        produced = sb.step_device(d.data_ptr(), 4000, scores.data_ptr())
        torch.cuda.synchronize()
        got = scores.cpu().numpy()
        for s in range(S):
            rc, p, want = ocs[s].step(sl[s])
            assert rc == 0 and p == produced, (k, s)
            if p:
                assert (bits(got[s]) == bits(want)).all(), (k, s)
        n_produced += int(produced)
    assert n_produced == n_steps - 3
    sb.close()
    gm.close()


def test_mfe_model_sdk_entry_points_in_a_fresh_process():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "mfe_sdk_worker.py")], stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                         text=True, timeout=600)
    assert out.returncode == 0 and "OK" in out.stdout, out.stdout[-2000:] + out.stderr[-3000:]
