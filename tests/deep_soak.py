"""Opt-in deep parity sweep (not collected by pytest; run on the GPU box):  python tests/deep_soak.py [rounds] [clips]

Every clip of `rounds` full batches (default 2 x 65 536) goes through the HIP path AND the CPU restatement of the reference
(oracle/, one worker process per host core), for the shipped int8 impulse and for the 49x40 fp32 headline graph: MFCC
features compared bit for bit, int8 input tensors and int8-model scores exactly, float-model scores within 1e-6.
Then 3 000 windows per model go through run_classifier() one by one (the latency-mode kernel) and must equal the batch path's
scores bit for bit, and 4 096 continuous-mode streams advance 14 slices with every 16th stream followed by the restated
run_classifier_continuous().  Writes one summary line per model and check; exit status 1 on any difference.

  python tests/deep_soak.py [rounds] [clips] fast   -- the same sweep for KWS_MODE_FAST against its tolerance (fast_sweep below)
"""
import multiprocessing as mp
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from kws_testlib import MODELS, ROOT, Oracle, OracleModel  # noqa: E402

_W = {}


def _worker(args):
    path, seed, first, n = args
    if path not in _W:
        o = _W.setdefault("oracle", Oracle())
        _W[path] = OracleModel(o, path)
    om, o = _W[path], _W["oracle"]
    clips = o.synth(seed, first, n)
    s, f, q = om.run_batch(clips, want_features=True)
    return first, s, f, q


def fast_sweep(rounds, B):
    """KWS_MODE_FAST over `rounds` full batches per model, every clip against the oracle: float32 graphs -- the maximum score
    difference (bar: 1e-4); int8 graphs -- clips whose scores changed, all of which must be explained by a flipped int8 input
    value (same tensor => same scores, bit for bit)."""
    sys.path.insert(0, ROOT)
    import torch
    from __graft_entry__ import load_package
    pkg = load_package()
    cores = len(os.sched_getaffinity(0))
    chunk, bad = 512, 0
    with mp.get_context("spawn").Pool(cores) as pool:
        for name in ("cfg2_mfcc40_f32.kwsm", "l476_no_yes_f32.kwsm", "cfg5_dscnn_mfcc40_f32.kwsm", "l476_no_yes.kwsm", "cfg2_mfcc40_int8.kwsm"):
            path = os.path.join(MODELS, name)
            gm = pkg.Model(path, device=0)
            gm.set_mode(pkg.MODE_FAST)
            F, C = gm.n_features, gm.n_labels
            t0 = time.time()
            n_clips = n_over = n_changed = n_unexplained = n_flips = n_back = 0
            max_score = max_feat = 0.0
            for r in range(rounds):
                seed, base = 2000 + r, 11 * r * B
                pcm = torch.empty((B, 16000), dtype=torch.int16, device="cuda:0")
                pkg.synth_clips_device(seed, base, B, 16000, pcm.data_ptr())
                feats = torch.empty((B, F), dtype=torch.float32, device="cuda:0")
                scores = torch.empty((B, C), dtype=torch.float32, device="cuda:0")
                q = None if gm.is_float else torch.empty((B, F), dtype=torch.int8, device="cuda:0")
                gm.run_classifier_batch_device(pcm.data_ptr(), B, scores.data_ptr(), None, None)           # the timed form: scores only
                torch.cuda.synchronize()
                gs = scores.cpu().numpy()
                n_back += gm.fast_fallback_count()
                gm.run_classifier_batch_device(pcm.data_ptr(), B, scores.data_ptr(), feats.data_ptr(), None if q is None else q.data_ptr())
                torch.cuda.synchronize()
                gf = feats.cpu().numpy()
                gq = None if q is None else q.cpu().numpy()
                jobs = [(path, seed, base + i, min(chunk, B - i)) for i in range(0, B, chunk)]
                for first, so, fo, qo in pool.imap_unordered(_worker, jobs):
                    i = first - base
                    n = len(so)
                    d = np.abs(gs[i:i + n] - so).max(axis=1)
                    max_feat = max(max_feat, float(np.abs(gf[i:i + n] - fo).max()))
                    if gm.is_float:
                        max_score = max(max_score, float(d.max()))
                        n_over += int((d > 1e-4).sum())
                    else:
                        flips = (gq[i:i + n] != qo).sum(axis=1)
                        n_flips += int(flips.sum())
                        n_changed += int((d > 0).sum())
                        n_unexplained += int(((d > 0) & (flips == 0)).sum())
                    n_clips += n
            ok = (n_over == 0) if gm.is_float else (n_unexplained == 0 and n_changed <= 0.02 * n_clips)
            bad += 0 if ok else 1
            if gm.is_float:
                print("%s KWS_MODE_FAST (fused network: %s): %d clips, max |score - oracle| = %.3g, %d clips over 1e-4, max |feature - oracle| = %.3g, "
                      "%d clips handed back to the exact kernels; %s (%.0f s)" % (name, gm.fast_is_fused, n_clips, max_score, n_over, max_feat, n_back,
                                                                                "OK" if ok else "MISMATCH", time.time() - t0), flush=True)
            else:
                print("%s KWS_MODE_FAST: %d clips, %.4f int8 input values per clip on the other side of a rounding boundary, %d clips with a changed "
                      "score (%d of them with an identical input tensor), max |feature - oracle| = %.3g, %d clips handed back; %s (%.0f s)"
                      % (name, n_clips, n_flips / max(1, n_clips), n_changed, n_unexplained, max_feat, n_back, "OK" if ok else "MISMATCH",
                         time.time() - t0), flush=True)
            gm.close()
    sys.exit(1 if bad else 0)


def main():
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
    if len(sys.argv) > 3 and sys.argv[3] == "fast":
        return fast_sweep(rounds, B)
    sys.path.insert(0, ROOT)
    import torch
    from __graft_entry__ import load_package
    pkg = load_package()
    cores = len(os.sched_getaffinity(0))
    chunk = 512
    bad = 0
    with mp.get_context("spawn").Pool(cores) as pool:
        for name in os.environ.get("KWS_SOAK_MODELS", "l476_no_yes.kwsm,cfg2_mfcc40_f32.kwsm").split(","):
            path = os.path.join(MODELS, name)
            gm = pkg.Model(path, device=0)
            F, C = gm.n_features, gm.n_labels
            t0 = time.time()
            n_feat_diff = n_q_diff = n_clips = 0
            max_score = 0.0
            for r in range(rounds):
                seed, base = 1000 + r, 7 * r * B
                pcm = torch.empty((B, 16000), dtype=torch.int16, device="cuda:0")
                pkg.synth_clips_device(seed, base, B, 16000, pcm.data_ptr())
                feats = torch.empty((B, F), dtype=torch.float32, device="cuda:0")
                scores = torch.empty((B, C), dtype=torch.float32, device="cuda:0")
                q = None if gm.is_float else torch.empty((B, F), dtype=torch.int8, device="cuda:0")
                gm.run_classifier_batch_device(pcm.data_ptr(), B, scores.data_ptr(), feats.data_ptr(), None if q is None else q.data_ptr())
                torch.cuda.synchronize()
                gf, gs = feats.cpu().numpy(), scores.cpu().numpy()
                gq = None if q is None else q.cpu().numpy()
                jobs = [(path, seed, base + i, min(chunk, B - i)) for i in range(0, B, chunk)]
                for first, s, f, qq in pool.imap_unordered(_worker, jobs):
                    i = first - base
                    n = len(s)
                    n_feat_diff += int((gf[i:i + n].view(np.uint32) != f.view(np.uint32)).sum())
                    if gq is not None:
                        n_q_diff += int((gq[i:i + n] != qq).sum())
                        max_score = max(max_score, float(np.abs(gs[i:i + n] - s).max()))
                        n_q_diff += int((gs[i:i + n].view(np.uint32) != s.view(np.uint32)).sum())
                    else:
                        max_score = max(max_score, float(np.abs(gs[i:i + n] - s).max()))
                    n_clips += n
            ok = n_feat_diff == 0 and n_q_diff == 0 and max_score <= (1e-6 if gm.is_float else 0.0)
            bad += 0 if ok else 1
            print("%s: %d clips, %d feature words compared: %d differ; int8 tensor/score words differing: %d; max |score - oracle| = %.3g; "
                  "%s (%.0f s, %d oracle workers)" % (name, n_clips, n_clips * F, n_feat_diff, n_q_diff, max_score,
                                                      "OK" if ok else "MISMATCH", time.time() - t0, cores), flush=True)
            gm.close()
    if "KWS_SOAK_MODELS" in os.environ:                    # other models: the batch sweep only
        sys.exit(1 if bad else 0)
    # the latency-mode kernel (run_classifier(), one window per call) against the batch path, many windows
    import ctypes
    o = Oracle()
    n_lat = 3000
    for name in ("l476_no_yes.kwsm", "cfg2_mfcc40_f32.kwsm"):
        gm = pkg.Model(os.path.join(MODELS, name), device=0)
        gm.set_default()
        clips = o.synth(4242, 10 ** 6, n_lat)
        want = gm.run_classifier_batch(clips)
        res = pkg.result_struct(gm.n_labels)()
        diff = 0
        for ci in range(n_lat):
            buf = clips[ci].astype(np.float32) / np.float32(32768)

            @pkg.GET_DATA_FN
            def get_data(offset, length, out):
                ctypes.memmove(out, buf[offset:offset + length].ctypes.data, 4 * length)
                return 0
            sig = pkg.Signal(get_data=get_data, total_length=16000)
            rc = pkg.lib().run_classifier(ctypes.byref(sig), ctypes.byref(res), False)
            got = np.float32([res.classification[i].value for i in range(gm.n_labels)])
            diff += int(rc != 0 or (got.view(np.uint32) != want[ci].view(np.uint32)).any())
        print("%s: run_classifier() (latency-mode kernel) on %d windows: %d differ from the batch path" % (name, n_lat, diff), flush=True)
        bad += 1 if diff else 0
        gm.close()
    # continuous mode: 4096 streams in lock step (state in HBM) for 14 slices; every 16th stream is followed by the restated
    # run_classifier_continuous() slice by slice
    from kws_testlib import OracleContinuous
    S, steps = 4096, 14
    gm = pkg.Model(os.path.join(MODELS, "l476_no_yes.kwsm"), device=0)
    om = OracleModel(o, os.path.join(MODELS, "l476_no_yes.kwsm"))
    audio = o.synth(555, 0, S * 4).reshape(S, 4 * 16000)[:, :steps * 4000]
    d_audio = torch.from_numpy(np.ascontiguousarray(audio)).to("cuda:0")
    sb = pkg.StreamBatch(gm, S)
    watch = list(range(0, S, 16))
    ocs = {i: OracleContinuous(om) for i in watch}
    for oc in ocs.values():
        oc.init()
    scores = torch.empty((S, gm.n_labels), dtype=torch.float32, device="cuda:0")
    diff = 0
    for k in range(steps):
        sl = d_audio[:, k * 4000:(k + 1) * 4000].contiguous()
        produced = sb.step_device(sl.data_ptr(), 4000, scores.data_ptr())
        torch.cuda.synchronize()
        got = scores.cpu().numpy()
        for i in watch:
            rc, p_, want = ocs[i].step(audio[i, k * 4000:(k + 1) * 4000])
            diff += int(rc != 0 or p_ != produced or (p_ and (got[i].view(np.uint32) != want.view(np.uint32)).any()))
    print("l476_no_yes.kwsm: %d streams x %d slices in continuous mode, %d streams followed by the oracle: %d differing steps" % (S, steps, len(watch), diff), flush=True)
    bad += 1 if diff else 0
    sb.close()
    gm.close()
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
