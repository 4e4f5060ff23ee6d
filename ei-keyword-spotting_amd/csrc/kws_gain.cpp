// kws_gain.cpp -- how far the loaded float32 graph's logits move per unit of feature error: the model-dependent half of
// KWS_MODE_FAST's guard (kws_fast_plan.cpp: build_guard).
//
// The fast kernel's features differ from the reference's by small, sign-random errors d[r][c] (row = frame, column = cepstral
// coefficient).  A logit DIFFERENCE z_a - z_b -- what the SOFTMAX sees -- then moves by sum_{r,c} (J_a - J_b)[r][c] d[r][c], J = d logits /
// d features; for independent errors its variance is sum (J_a - J_b)^2 d^2.  cmvnw's windows overlap in all but one row, so a column's
// error level is nearly the same in every row, and the column's share of that variance is
//     col_gain[c]^2 x sum_r d[r][c]^2,      col_gain[c]^2 = max over label pairs (a, b) of  sum_r (J_a - J_b)[r][c]^2 / n_frames.
// J depends on the input through the ReLU / clamp / max-pool pattern, so it is evaluated -- exactly, by reverse differentiation of the
// reference's float kernels (TFL/kernels/internal/reference/conv.h:28-99, depthwiseconv_float.h:25, add.h:179-215, pooling.h:189-237,
// fully_connected.h:26-60) in double -- on a calibration set and the LARGEST value per column is kept.  The network's input is cmvnw's
// output: every column standardised over its window, so the calibration set is N matrices of unit-variance Gaussian columns with a
// first-order correlation along time drawn per matrix (0 .. 0.95: white to slowly varying).  A CALIBRATED ESTIMATE, not a bound (a
// bound through |W| row sums is ~100x larger and would switch the fast mode off for every model): tests/test_gain_calibration.py
// holds it to Jacobians of the shipped graphs on real clips' features, and the guard multiplies it by k standard deviations.
//
// sigma_net: what the fused network's own arithmetic (matrix-core accumulation of split 22-bit operands instead of the reference's
// sequential total += x * w) moves in a logit difference -- the same graph evaluated in float32 both ways on the calibration set.
#include "kws_internal.h"

#include <random>

namespace {

struct BlockTrace {
    std::vector<double> out;        // [pool_w][out_c] what the next block reads
    std::vector<double> dmask;      // [out_w][out_c] d(value after the clamps) / d(accumulator): 0 or 1
    std::vector<int> arg;           // [pool_w][out_c] row of the window's maximum
    std::vector<double> pmask;      // [pool_w][out_c] d(after the pooling clamp) / d(maximum)
};

inline double clampd(double v, double lo, double hi, double *inside)
{
    *inside = (v > lo && v < hi) ? 1.0 : 0.0;
    return v < lo ? lo : v > hi ? hi : v;
}

// one block forward in double; in [in_w][in_c]
void block_forward(const KwsConvBlockF32 &k, const std::vector<float> &w, const std::vector<float> &bias, const std::vector<float> &addc,
                   const std::vector<double> &in, BlockTrace &t)
{
    const int out_w = k.out_w, out_c = k.out_c, in_c = k.in_c, taps = k.taps;
    std::vector<double> val((size_t)out_w * out_c);
    t.dmask.assign((size_t)out_w * out_c, 0.0);
    for (int r = 0; r < out_w; r++)
        for (int n = 0; n < out_c; n++) {
            double s = 0.0;
            for (int tap = 0; tap < taps; tap++) {
                const int row = r + tap - k.pad_left;
                if (row < 0 || row >= k.in_w) continue;
                if (k.depthwise) s += in[(size_t)row * in_c + n / k.depth_mult] * (double)w[(size_t)tap * out_c + n];
                else {
                    const float *wr = &w[((size_t)n * taps + tap) * in_c];
                    const double *xr = &in[(size_t)row * in_c];
                    for (int c = 0; c < in_c; c++) s += xr[c] * (double)wr[c];
                }
            }
            double m1, m2 = 1.0;
            double v = clampd(s + (double)bias[n], k.conv_min, k.conv_max, &m1);
            if (k.has_add) v = clampd(v + (double)addc[n], k.add_min, k.add_max, &m2);
            val[(size_t)r * out_c + n] = v;
            t.dmask[(size_t)r * out_c + n] = m1 * m2;
        }
    const int pw = k.pool_w;
    t.out.assign((size_t)pw * out_c, 0.0);
    t.arg.assign((size_t)pw * out_c, 0);
    t.pmask.assign((size_t)pw * out_c, 1.0);
    for (int j = 0; j < pw; j++)
        for (int n = 0; n < out_c; n++) {
            int best = std::min(j * k.pool_stride, out_w - 1);
            double m = val[(size_t)best * out_c + n];
            for (int i = 1; i < k.pool; i++) {
                const int r = j * k.pool_stride + i;
                if (r >= out_w) break;                                  // ragged last window: clipped (pooling.h:206-214)
                if (val[(size_t)r * out_c + n] > m) { m = val[(size_t)r * out_c + n]; best = r; }
            }
            double pm;
            t.out[(size_t)j * out_c + n] = clampd(m, k.pool_min, k.pool_max, &pm);
            t.arg[(size_t)j * out_c + n] = best;
            t.pmask[(size_t)j * out_c + n] = pm;
        }
}

// gradients of L seeds through one block: g_out [L][pool_w][out_c] -> g_in [L][in_w][in_c]
void block_backward(const KwsConvBlockF32 &k, const std::vector<float> &w, const BlockTrace &t, int L, const std::vector<double> &g_out,
                    std::vector<double> &g_in)
{
    const int out_w = k.out_w, out_c = k.out_c, in_c = k.in_c, taps = k.taps, pw = k.pool_w;
    g_in.assign((size_t)L * k.in_w * in_c, 0.0);
    std::vector<double> gs((size_t)out_w * out_c);
    for (int l = 0; l < L; l++) {
        std::fill(gs.begin(), gs.end(), 0.0);
        const double *go = &g_out[(size_t)l * pw * out_c];
        for (int j = 0; j < pw; j++)
            for (int n = 0; n < out_c; n++) {
                const size_t o = (size_t)j * out_c + n;
                gs[(size_t)t.arg[o] * out_c + n] += go[o] * t.pmask[o];
            }
        double *gi = &g_in[(size_t)l * k.in_w * in_c];
        for (int r = 0; r < out_w; r++)
            for (int n = 0; n < out_c; n++) {
                const double g = gs[(size_t)r * out_c + n] * t.dmask[(size_t)r * out_c + n];
                if (g == 0.0) continue;
                for (int tap = 0; tap < taps; tap++) {
                    const int row = r + tap - k.pad_left;
                    if (row < 0 || row >= k.in_w) continue;
                    if (k.depthwise) gi[(size_t)row * in_c + n / k.depth_mult] += g * (double)w[(size_t)tap * out_c + n];
                    else {
                        const float *wr = &w[((size_t)n * taps + tap) * in_c];
                        double *gr = &gi[(size_t)row * in_c];
                        for (int c = 0; c < in_c; c++) gr[c] += g * (double)wr[c];
                    }
                }
            }
    }
}

// the graph in float32 with the reference's summation order (blocked = false: total += x * w, tap outer, channel inner) or with partial
// sums over groups of four products added pairwise (blocked = true: the shape of a matrix-core k-step); logits only
// x s as a 22-bit number: the sum of the two halves the split-operand contraction carries (kws_fast.hip: fast_split_image; kws_fast_plan.cpp
// splits the weights the same way); s = 2^14 / 2^ceil(log2 max|v|)
float half_rn(float v)
{
    if (!(fabsf(v) < 65520.0f)) return v;
    if (fabsf(v) < 6.103515625e-5f) return ldexpf(nearbyintf(ldexpf(v, 24)), -24);       // subnormal halves: multiples of 2^-24
    int e;
    const float m = frexpf(v, &e);                                                      // v = m 2^e, 0.5 <= |m| < 1: 11 bits = multiples of 2^-11
    return ldexpf(nearbyintf(ldexpf(m, 11)), e - 11);
}
void split22(const std::vector<float> &v, std::vector<float> &out, float *scale)
{
    float mx = 0.0f;
    for (float a : v) mx = std::max(mx, fabsf(a));
    int e = 0;
    if (mx > 0.0f && std::isfinite(mx)) (void)frexpf(mx, &e);
    e = std::max(-100, std::min(100, e));
    const float s = ldexpf(1.0f, 14 - e);
    out.resize(v.size());
    for (size_t i = 0; i < v.size(); i++) {
        const float y = v[i] * s, hi = half_rn(y), lo = half_rn(y - hi);
        out[i] = hi + lo;                                                               // exact in fp32: 22 significant bits
    }
    *scale = ldexpf(1.0f, e - 14);
}

void forward_f32(const KwsNnPlanF32 &N, const kws_handle::HostF32 &W, const std::vector<float> &x, bool blocked, std::vector<float> &logits)
{
    std::vector<float> cur = x, nxt, val, cs, ws;
    for (int b = 0; b < N.n_blocks; b++) {
        const KwsConvBlockF32 &k = N.blk[b];
        const std::vector<float> *wp = &W.w[b];
        float unscale = 1.0f;
        if (blocked && kws_fast_block_splits(k)) {
            // the fused kernel's contraction of exactly the blocks it runs that way (one predicate with the plan builder): both operands as 22-bit
            // numbers (the dropped lo x lo term is below the accumulator's own rounding).  The partial sums below are four products deep where the
            // matrix instruction is thirty-two in three passes: sigma_net is an estimate of the ORDER of the network's own noise (it enters V next to
            // terms ten times its size), not a model of the instruction
            float s1, s2;
            split22(cur, cs, &s1);
            split22(W.w[b], ws, &s2);
            cur.swap(cs);
            wp = &ws;
            unscale = s1 * s2;
        }
        const std::vector<float> &w = *wp;
        val.assign((size_t)k.out_w * k.out_c, 0.0f);
        for (int r = 0; r < k.out_w; r++)
            for (int n = 0; n < k.out_c; n++) {
                float total = 0.0f, part[4] = { 0.f, 0.f, 0.f, 0.f };
                int cnt = 0;
                auto add = [&](float a, float c) {
                    if (!blocked) { const float p = a * c; total = total + p; return; }
                    part[cnt & 3] = a * c;
                    if ((++cnt & 3) == 0) { total = total + ((part[0] + part[1]) + (part[2] + part[3])); part[0] = part[1] = part[2] = part[3] = 0.f; }
                };
                for (int tap = 0; tap < k.taps; tap++) {
                    const int row = r + tap - k.pad_left;
                    if (row < 0 || row >= k.in_w) continue;
                    if (k.depthwise) add(cur[(size_t)row * k.in_c + n / k.depth_mult], w[(size_t)tap * k.out_c + n]);
                    else for (int c = 0; c < k.in_c; c++) add(cur[(size_t)row * k.in_c + c], w[((size_t)n * k.taps + tap) * k.in_c + c]);
                }
                if (blocked && (cnt & 3)) total = total + ((part[0] + part[1]) + (part[2] + part[3]));
                float v = total * unscale + W.bias[b][n];
                v = std::min(std::max(v, k.conv_min), k.conv_max);
                if (k.has_add) { v = v + W.addc[b][n]; v = std::min(std::max(v, k.add_min), k.add_max); }
                val[(size_t)r * k.out_c + n] = v;
            }
        nxt.assign((size_t)k.pool_w * k.out_c, 0.0f);
        for (int j = 0; j < k.pool_w; j++)
            for (int n = 0; n < k.out_c; n++) {
                float m = val[(size_t)std::min(j * k.pool_stride, k.out_w - 1) * k.out_c + n];
                for (int i = 1; i < k.pool && j * k.pool_stride + i < k.out_w; i++) m = std::max(m, val[(size_t)(j * k.pool_stride + i) * k.out_c + n]);
                nxt[(size_t)j * k.out_c + n] = std::min(std::max(m, k.pool_min), k.pool_max);
            }
        cur.swap(nxt);
    }
    logits.assign((size_t)N.fc_out, 0.0f);
    for (int o = 0; o < N.fc_out; o++) {
        float total = 0.0f, t2 = 0.0f;
        for (int i = 0; i < N.fc_in; i++) {
            const float p = cur[(size_t)i] * W.fc_w[(size_t)o * N.fc_in + i];
            if (blocked && (i & 1)) t2 = t2 + p; else total = total + p;
        }
        total = (total + t2) + W.fc_b[(size_t)o];
        logits[(size_t)o] = std::min(std::max(total, N.fc_min), N.fc_max);
    }
}

}   // namespace

// Fills h->gain from h->nnf / h->hostf (float32 graphs).  Never fails the model: a graph whose gains come out non-finite gets
// infinite gains, i.e. every clip of a fast-mode call is handed to the exact kernels.
void kws_calibrate_gain(kws_handle *h)
{
    kws_handle::Gain &G = h->gain;
    const KwsNnPlanF32 &N = h->nnf;
    const kws_handle::HostF32 &W = h->hostf;
    const int nfr = N.blk[0].in_w, ncep = N.blk[0].in_c, L = N.fc_out, F = nfr * ncep;
    G.col.assign((size_t)ncep, 0.0f);
    G.sigma_net = 0.0f;
    G.n_inputs = KWS_GAIN_INPUTS;
    G.calibrated = 1;
    std::mt19937 rng(0x6b7773u);                            // fixed: the same model always gets the same guard
    auto uniform = [&]() { return ((double)(rng() >> 5) + 0.5) / 134217728.0; };        // (0, 1), 27 bits
    auto normal = [&]() { return sqrt(-2.0 * log(uniform())) * cos(2.0 * M_PI * uniform()); };
    static const double kRho[4] = { 0.0, 0.5, 0.8, 0.95 };
    std::vector<BlockTrace> tr((size_t)N.n_blocks);
    std::vector<double> colmax((size_t)ncep, 0.0), x((size_t)F), g, g2;
    std::vector<float> xf((size_t)F), z0, z1;
    double net2 = 0.0;
    long net_n = 0;
    for (int it = 0; it < KWS_GAIN_INPUTS; it++) {
        const double rho = kRho[it & 3], nz = sqrt(1.0 - rho * rho);
        for (int c = 0; c < ncep; c++) {
            double v = normal();
            for (int r = 0; r < nfr; r++) { x[(size_t)r * ncep + c] = v; v = rho * v + nz * normal(); }
        }
        // forward, keeping what the reverse pass needs
        const std::vector<double> *cur = &x;
        for (int b = 0; b < N.n_blocks; b++) { block_forward(N.blk[b], W.w[b], W.bias[b], W.addc[b], *cur, tr[(size_t)b]); cur = &tr[(size_t)b].out; }
        // seeds: one per logit, through the FULLY_CONNECTED layer's clamp
        g.assign((size_t)L * N.fc_in, 0.0);
        for (int o = 0; o < L; o++) {
            double zz = (double)W.fc_b[(size_t)o];
            for (int i = 0; i < N.fc_in; i++) zz += (*cur)[(size_t)i] * (double)W.fc_w[(size_t)o * N.fc_in + i];
            double inside;
            (void)clampd(zz, N.fc_min, N.fc_max, &inside);
            if (inside != 0.0) for (int i = 0; i < N.fc_in; i++) g[(size_t)o * N.fc_in + i] = (double)W.fc_w[(size_t)o * N.fc_in + i];
        }
        for (int b = N.n_blocks - 1; b >= 0; b--) { block_backward(N.blk[b], W.w[b], tr[(size_t)b], L, g, g2); g.swap(g2); }
        // g = J [L][n_frames][n_cepstral]; per column: the largest label pair's sum over the rows of (J_a - J_b)^2
        for (int c = 0; c < ncep; c++) {
            double best = 0.0;
            for (int a = 0; a < L; a++)
                for (int b2 = a + 1; b2 < L; b2++) {
                    double s = 0.0;
                    for (int r = 0; r < nfr; r++) {
                        const double d = g[((size_t)a * nfr + r) * ncep + c] - g[((size_t)b2 * nfr + r) * ncep + c];
                        s += d * d;
                    }
                    best = std::max(best, s);
                }
            colmax[(size_t)c] = std::max(colmax[(size_t)c], best);
        }
        // the network's own re-ordering noise
        for (int i = 0; i < F; i++) xf[(size_t)i] = (float)x[(size_t)i];
        forward_f32(N, W, xf, false, z0);
        forward_f32(N, W, xf, true, z1);
        for (int a = 0; a < L; a++)
            for (int b2 = a + 1; b2 < L; b2++) {
                const double d = ((double)z1[(size_t)a] - (double)z0[(size_t)a]) - ((double)z1[(size_t)b2] - (double)z0[(size_t)b2]);
                net2 += d * d;
                net_n++;
            }
    }
    double tot = 0.0;
    bool finite = true;
    // Headroom: the largest value over a finite calibration set underestimates what a real clip's ReLU / max-pool pattern can
    // reach -- measured on real clips' Jacobians (tests/test_gain_calibration.py): a column up to 1.7 x, a clip's total gain up to 1.2 x the
    // plain maximum over 16 matrices.  48 matrices and a factor KWS_GAIN_HEADROOM on every column put the total gain of every measured clip below
    // the calibrated one and a single column at most ~1.3 x above: k_sigma = 4.5 against the calibrated gain is then >= 3.4 sigma for a clip
    // whose error sat in its worst column alone (kws.h states this next to k_sigma).
    for (int c = 0; c < ncep; c++) {
        const double v = (double)KWS_GAIN_HEADROOM * sqrt(colmax[(size_t)c] / (double)nfr);
        if (!std::isfinite(v)) finite = false;
        G.col[(size_t)c] = (float)v;
        tot += colmax[(size_t)c];
    }
    G.total = (float)((double)KWS_GAIN_HEADROOM * sqrt(tot));
    G.sigma_net = net_n ? (float)(2.0 * sqrt(net2 / (double)net_n)) : 0.0f;      // twice the measured rms: two orders are one sample of the spread
    if (!finite || !std::isfinite(G.total) || !std::isfinite(G.sigma_net)) {
        for (float &v : G.col) v = INFINITY;
        G.total = INFINITY;
        G.sigma_net = INFINITY;
    }
}
