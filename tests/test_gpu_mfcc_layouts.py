"""-m gpu: the two spectral layouts of the exact MFCC path.  kws_mfcc8_kernel (eight lanes per frame, eight frames per pass, a tail pass
for a remainder of one or two frames) serves the int16 batch calls; kws_mfcc_kernel (32 lanes per frame, frame pairs) keeps the float
samples and stays selectable through the development switch KWS_DEV_MFCC_OLD_LAYOUT.  Every frame count the tuned kernels accept is a
different mix of full passes, partial passes and tail passes: for each the features (one launch: spectral + cmvnw), the cepstra before
cmvnw and the mel / frame energies must equal the oracle's bit for bit, and the two layouts must equal each other on a larger batch."""
import os

import numpy as np
import pytest

from kws_testlib import ROOT, OracleModel, bits, special_clips, synth_model_blob

pytestmark = pytest.mark.gpu

# samples per window -> frames = floor((n - 320) / 320): 1, 2 (no tail: fewer than 8), 7, 8, 9 / 10 (8 + tail), 11 (partial pass),
# 16, 17, 24, 25, 26, 33, 41, 48, 49, 50 and -- 32 filters only -- 52
FRAMES = (1, 2, 7, 8, 9, 10, 11, 16, 17, 24, 25, 26, 33, 41, 48, 49, 50, 52)
SHAPES = {"32x13": dict(num_filters=32, ncep=13), "40x40": dict(num_filters=40, ncep=40, high=0), "40x13": dict(num_filters=40, ncep=13, high=0)}


@pytest.fixture(scope="module")
def pkg():
    import sys
    sys.path.insert(0, ROOT)
    import torch  # noqa: F401
    from __graft_entry__ import load_package
    return load_package()


@pytest.mark.parametrize("shape", sorted(SHAPES))
def test_every_frame_count_on_both_layouts(shape, dev_pkg, oracle, tmp_path):
    import torch
    pkg = dev_pkg            # the layout switches are development switches: the development build of the library (conftest.py)
    ran = []
    for nfr in FRAMES:
        if nfr > 50 and SHAPES[shape]["num_filters"] == 40:
            continue
        n = 320 * (nfr + 1) + (8 if nfr % 3 == 0 else 0)            # (some windows with samples beyond the last frame; rows stay 16-byte aligned)
        blob = synth_model_blob(seed=nfr, raw_samples=n, blocks=((8, 3, 1), (4, 3, 1)), n_labels=3, **SHAPES[shape])
        path = str(tmp_path / ("m%d.kwsm" % nfr))
        open(path, "wb").write(blob)
        om = OracleModel(oracle, path)
        gm = pkg.Model(blob=blob)
        assert gm.n_frames == nfr and om.raw_sample_count == n
        if gm.mfcc_kernel in ("kws_spectral_lds_kernel", "kws_spectral_generic_kernel", "kws_mfcc8_kernel (chunked)"):
            gm.close()
            continue                                                 # a shape the tuned kernels leave to the general ones
        assert gm.mfcc_kernel == ("kws_mfcc8_kernel" if nfr >= 16 else "kws_mfcc_kernel")     # (short windows stay on the old layout by default)
        sp = special_clips()
        host = np.concatenate([oracle.synth(nfr, 0, 61, n), np.stack([np.resize(sp[k], n) for k in ("impulses", "zeros", "alternating_fullscale")])])
        B = len(host)
        d = torch.from_numpy(np.ascontiguousarray(host)).to("cuda:0")
        nf, F = gm.n_filters, gm.n_features
        res = {}
        for old in (False, True):
            os.environ["KWS_DEV_MFCC8_MIN_FRAMES"] = "1"            # the new layout for every frame count ...
            if old:
                os.environ["KWS_DEV_MFCC_OLD_LAYOUT"] = "1"          # ... unless the old one is forced
            try:
                feat = torch.zeros((B, F), dtype=torch.float32, device="cuda:0")
                q = torch.zeros((B, F), dtype=torch.int8, device="cuda:0")
                cep = torch.zeros((B, F), dtype=torch.float32, device="cuda:0")
                mel = torch.zeros((B, nfr * nf), dtype=torch.float32, device="cuda:0")
                en = torch.zeros((B, nfr), dtype=torch.float32, device="cuda:0")
                gm.extract_mfcc_batch_device(d.data_ptr(), B, feat.data_ptr(), q.data_ptr())
                gm.mfcc_batch_device(d.data_ptr(), B, cep.data_ptr())
                gm.mfe_batch_device(d.data_ptr(), B, mel.data_ptr(), en.data_ptr())
                torch.cuda.synchronize()
            finally:
                os.environ.pop("KWS_DEV_MFCC_OLD_LAYOUT", None)
                os.environ.pop("KWS_DEV_MFCC8_MIN_FRAMES", None)
            res[old] = [x.cpu().numpy() for x in (feat, q, cep, mel, en)]
        for a, b in zip(res[False], res[True]):
            assert (bits(a) == bits(b)).all(), (shape, nfr)
        feat, q, cep, mel, en = res[False]
        so, fo, qo = om.run_batch(host, want_features=True)
        assert (bits(feat) == bits(fo)).all() and (q == qo).all(), (shape, nfr)
        for i in (0, 1, B - 3, B - 2, B - 1):
            assert (bits(cep[i]) == bits(oracle.mfcc_nocmvn(host[i], om.cfg).reshape(-1))).all(), (shape, nfr, i)
            mo, eo = oracle.mfe(host[i], om.cfg)
            assert (bits(mel[i]) == bits(mo.reshape(-1))).all() and (bits(en[i]) == bits(eo.reshape(-1))).all(), (shape, nfr, i)
        ran.append(nfr)
        gm.close()
    print(shape, "frame counts on the tuned kernels:", ran)
    assert len(ran) >= len(FRAMES) - 2, ran
