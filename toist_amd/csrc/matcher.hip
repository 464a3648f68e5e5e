// Hungarian matcher for the TOIST/MDETR set criterion -- one workgroup per (decoder layer, image).
//
// Replaces /root/reference/models/matcher.py:60-87 (cost build on device, `.cpu()` sync, per-image
// scipy.optimize.linear_sum_assignment on the host).  Here the [Q, T_i] cost block is built in
// fp32 in the reference's evaluation order (compiled with -ffp-contract=off so no FMA fusion
// changes a rounding), kept in LDS, and solved in fp64 by ONE wavefront with the same
// shortest-augmenting-path traversal and tie rules as SciPy's rectangular LSAP (restated in
// oracle/lsap.c).  All decoder layers x images go in one launch; indices stay on the device.
//
// LDS-resident, latency-bound: neither the HBM nor the MFMA roofline applies.
#include <mutex>

#include "common.h"

namespace toist {

struct PickKey {
    double val;
    int it;   // position in the `remaining` list
    int una;  // candidate column is unassigned
};

// Combine rule equivalent to SciPy's sequential scan over `remaining`:
// lower value wins; on equal value an unassigned column wins, the LAST such one in scan order;
// with no unassigned column among the ties, the FIRST one in scan order.
__device__ __forceinline__ PickKey pick_combine(PickKey a, PickKey b) {
    if (a.val < b.val) return a;
    if (b.val < a.val) return b;
    if (a.val != b.val) return (a.val != a.val) ? b : a;  // NaN guard (never expected)
    if (a.una && b.una) return (a.it > b.it) ? a : b;
    if (a.una) return a;
    if (b.una) return b;
    return (a.it < b.it) ? a : b;
}

__device__ __forceinline__ PickKey wave_pick(PickKey k) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        PickKey other;
        other.val = __shfl_xor(k.val, o, 64);
        other.it = __shfl_xor(k.it, o, 64);
        other.una = __shfl_xor(k.una, o, 64);
        k = pick_combine(k, other);
    }
    return k;
}

__device__ __forceinline__ void wave_lds_fence() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// status codes written per (layer,image)
enum { ST_OK = 0, ST_INVALID = 1, ST_INFEASIBLE = 2 };

// Rectangular LSAP (R <= C) on an LDS-resident fp32 cost block, solved in fp64 by ONE wavefront with SciPy's
// shortest-augmenting-path traversal and tie rules (oracle/lsap.c restates the same traversal).  Leaves the
// assignment in col4row[R] / row4col[C]; returns ST_OK or ST_INFEASIBLE.
__device__ __forceinline__ int lsap_solve_wave(const float* cw, const int R, const int C, double* du, double* dv, double* spc, int* path,
                                               int* row4col, int* remaining, int* col4row, int* in_sr, int* in_sc, const int lane) {
    for (int j = lane; j < C; j += 64) { dv[j] = 0.0; path[j] = -1; row4col[j] = -1; }
    for (int i = lane; i < R; i += 64) { du[i] = 0.0; col4row[i] = -1; }
    wave_lds_fence();

    int st = ST_OK;
    for (int cur = 0; cur < R; ++cur) {
        for (int t = lane; t < C; t += 64) { remaining[t] = C - 1 - t; spc[t] = INFINITY; in_sc[t] = 0; }
        for (int i = lane; i < R; i += 64) in_sr[i] = 0;
        wave_lds_fence();

        int live = C, sink = -1, i = cur;
        double floor_val = 0.0;
        while (sink < 0) {
            if (lane == 0) in_sr[i] = 1;
            const double ui = du[i];
            const float* crow = cw + (size_t)i * C;
            PickKey best;  // per-lane replay of the reference scan, starting from lowest = +inf
            best.val = INFINITY; best.it = 0x7fffffff; best.una = 0;
            for (int t = lane; t < live; t += 64) {
                const int j = remaining[t];
                const double r = ((floor_val + (double)crow[j]) - ui) - dv[j];
                double s = spc[j];
                if (r < s) { path[j] = i; spc[j] = r; s = r; }
                const int una = row4col[j] < 0;
                if (s < best.val || (s == best.val && una)) { best.val = s; best.it = t; best.una = una; }
            }
            best = wave_pick(best);
            floor_val = best.val;
            if (!(floor_val < INFINITY)) { st = ST_INFEASIBLE; break; }
            const int pick = best.it;
            const int j = remaining[pick];
            const int owner = row4col[j];
            if (owner < 0) sink = j; else i = owner;
            wave_lds_fence();
            if (lane == 0) {
                in_sc[j] = 1;
                remaining[pick] = remaining[live - 1];
            }
            --live;
            wave_lds_fence();
        }
        if (st != ST_OK) break;

        // dual variables
        if (lane == 0) du[cur] += floor_val;
        for (int r = lane; r < R; r += 64)
            if (in_sr[r] && r != cur) du[r] += floor_val - spc[col4row[r]];
        for (int j = lane; j < C; j += 64)
            if (in_sc[j]) dv[j] -= floor_val - spc[j];
        wave_lds_fence();
        // augment along the path (serial, short)
        if (lane == 0) {
            int j = sink;
            for (;;) {
                const int pi = path[j];
                row4col[j] = pi;
                const int prev = col4row[pi];
                col4row[pi] = j;
                j = prev;
                if (pi == cur) break;
            }
        }
        wave_lds_fence();
    }

    return st;
}

// ---- the same traversal with the column state in REGISTERS (round 6) ---------------------------------------------------------------------
// lsap_solve_wave keeps spc / dv / path / row4col / remaining in LDS and walks `remaining` by position: every scan step is five dependent LDS
// reads per candidate plus a six-stage shuffle of a 16-byte key -- ~1.2 ms for one 95 x 95 softkd problem, 5.9 ms of the 26 ms distillation step
// (profiles/r05_distill_timeline.txt).  Here lane l OWNS columns l, l + 64, ... (CPL of them): shortest path cost, dual variable, predecessor,
// owner row and the column's POSITION in SciPy's `remaining` list live in its registers.  The list itself is never materialised: removing the
// element at position p moves the element at position live - 1 to p, i.e. one compare-and-set per owned column.  A scan step is then one LDS
// read per owned column (the cost row), a DPP min-reduction of one double, and -- only when several columns tie at the minimum -- the
// reference's tie rule on positions (an unassigned column wins, the LAST such in scan order; otherwise the FIRST in scan order).  Same
// arithmetic, same order of floating-point operations per column, same tie rules: bit-identical assignments (tests/golden/lsap_kat.json,
// oracle/lsap.c on the step's own matrices in tests/test_gpu_distill_fullsize.py).
// min of two doubles that are never NaN: one v_min_f64 (fmin() would canonicalise both operands first: two more f64 operations per call)
__device__ __forceinline__ double min_f64(double a, double b) {
    double r;
    asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
template <int CTRL>
__device__ __forceinline__ double dpp_f64(double v) {     // row rotations: every lane has a source, no identity needed
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xf, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double wave_min_f64(double v) {
    v = min_f64(v, dpp_f64<0x121>(v));        // row_ror:1   (rotations inside each row of 16 lanes: after 1, 2, 4, 8 every lane holds its row's minimum)
    v = min_f64(v, dpp_f64<0x122>(v));        // row_ror:2
    v = min_f64(v, dpp_f64<0x124>(v));        // row_ror:4
    v = min_f64(v, dpp_f64<0x128>(v));        // row_ror:8
    const int lo = __double2loint(v), hi = __double2hiint(v);
    const double r0 = __hiloint2double(__builtin_amdgcn_readlane(hi, 0), __builtin_amdgcn_readlane(lo, 0));
    const double r1 = __hiloint2double(__builtin_amdgcn_readlane(hi, 16), __builtin_amdgcn_readlane(lo, 16));
    const double r2 = __hiloint2double(__builtin_amdgcn_readlane(hi, 32), __builtin_amdgcn_readlane(lo, 32));
    const double r3 = __hiloint2double(__builtin_amdgcn_readlane(hi, 48), __builtin_amdgcn_readlane(lo, 48));
    return min_f64(min_f64(r0, r1), min_f64(r2, r3));
}
__device__ __forceinline__ int wave_max_i32(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = max(v, __shfl_xor(v, o, 64));
    return v;
}

template <int CPL>
__device__ __forceinline__ int lsap_solve_wave_reg(const float* cw, const int R, const int C, double* du, double* spc_lds, int* col4row, int* in_sr,
                                                   const int lane) {
    double dv[CPL], spc[CPL];
    int path[CPL], r4c[CPL], pos[CPL];
    bool gone[CPL];                       // the column has left `remaining` in this augmentation (= SciPy's SC), or does not exist (j >= C)
#pragma unroll
    for (int c = 0; c < CPL; ++c) { dv[c] = 0.0; path[c] = -1; r4c[c] = -1; }
    for (int i = lane; i < R; i += 64) { du[i] = 0.0; col4row[i] = -1; }
    wave_lds_fence();

    for (int cur = 0; cur < R; ++cur) {
#pragma unroll
        for (int c = 0; c < CPL; ++c) {
            const int j = lane + 64 * c;
            spc[c] = INFINITY;
            pos[c] = C - 1 - j;           // `remaining` starts in descending column order
            gone[c] = j >= C;
        }
        for (int i = lane; i < R; i += 64) in_sr[i] = 0;
        wave_lds_fence();

        int live = C, sink = -1, i = cur;
        double floor_val = 0.0;
        while (sink < 0) {
            if (lane == 0) in_sr[i] = 1;
            const float* crow = cw + (size_t)i * C;
            float cf[CPL];                                    // the row's costs of the owned columns and the row's dual: ONE LDS round trip per step
#pragma unroll
            for (int c = 0; c < CPL; ++c) cf[c] = crow[min(lane + 64 * c, C - 1)];
            const double ui = du[i];
            double mine = INFINITY;
#pragma unroll
            for (int c = 0; c < CPL; ++c) {                   // predicated, no branches: a column that has left `remaining` keeps its state and offers +inf
                const double r = ((floor_val + (double)cf[c]) - ui) - dv[c];
                const bool upd = !gone[c] && r < spc[c];
                spc[c] = upd ? r : spc[c];
                path[c] = upd ? i : path[c];
                mine = min_f64(mine, gone[c] ? (double)INFINITY : spc[c]);
            }
            const double m = wave_min_f64(mine);
            if (!(m < INFINITY)) return ST_INFEASIBLE;
            floor_val = m;
            // which column: the ties at the minimum, resolved as the sequential scan over `remaining` resolves them
            unsigned long long tie[CPL];
            int ties = 0;
#pragma unroll
            for (int c = 0; c < CPL; ++c) {
                tie[c] = __ballot(!gone[c] && spc[c] == m);
                ties += __popcll(tie[c]);
            }
            int pc = 0, pl = 0;           // the picked column's register slot and lane (wave-uniform)
            if (ties == 1) {
#pragma unroll
                for (int c = 0; c < CPL; ++c)
                    if (tie[c]) { pc = c; pl = __ffsll((long long)tie[c]) - 1; }
            } else {
                unsigned long long anyu = 0;
#pragma unroll
                for (int c = 0; c < CPL; ++c) anyu |= __ballot(!gone[c] && spc[c] == m && r4c[c] < 0);
                int key = -1;             // unassigned ties: the largest position; none: the smallest position
#pragma unroll
                for (int c = 0; c < CPL; ++c) {
                    const bool t = !gone[c] && spc[c] == m;
                    if (anyu) { if (t && r4c[c] < 0) key = max(key, pos[c]); }
                    else if (t) key = max(key, 0x7fffffff - pos[c]);
                }
                const int best = wave_max_i32(key);
#pragma unroll
                for (int c = 0; c < CPL; ++c) {
                    const bool t = !gone[c] && spc[c] == m && (anyu ? (r4c[c] < 0 && pos[c] == best) : (0x7fffffff - pos[c] == best));
                    const unsigned long long b = __ballot(t);
                    if (b) { pc = c; pl = __ffsll((long long)b) - 1; }
                }
            }
            pc = __builtin_amdgcn_readfirstlane(pc);
            pl = __builtin_amdgcn_readfirstlane(pl);
            int ppos = 0, owner = -1;
#pragma unroll
            for (int c = 0; c < CPL; ++c)
                if (c == pc) { ppos = __builtin_amdgcn_readlane(pos[c], pl); owner = __builtin_amdgcn_readlane(r4c[c], pl); }
            // remaining[pick] = remaining[--live]
            --live;
#pragma unroll
            for (int c = 0; c < CPL; ++c) {
                if (!gone[c] && pos[c] == live) pos[c] = ppos;
                if (c == pc && lane == pl) gone[c] = true;
            }
            if (owner < 0) sink = pl + 64 * pc; else i = owner;
        }

        // dual variables (the columns' shortest path costs go through LDS once: rows look their column up)
#pragma unroll
        for (int c = 0; c < CPL; ++c)
            if (lane + 64 * c < C) spc_lds[lane + 64 * c] = spc[c];
        wave_lds_fence();
        if (lane == 0) du[cur] += floor_val;
        for (int r = lane; r < R; r += 64)
            if (in_sr[r] && r != cur) du[r] += floor_val - spc_lds[col4row[r]];
#pragma unroll
        for (int c = 0; c < CPL; ++c)
            if (gone[c] && lane + 64 * c < C) dv[c] -= floor_val - spc[c];
        wave_lds_fence();
        // augment along the path (serial, short): every lane follows it, the owning lane records the new row of each column
        int j = sink;
        for (;;) {
            const int jc = j >> 6, jl = j & 63;
            int pi = -1;
#pragma unroll
            for (int c = 0; c < CPL; ++c)
                if (c == jc) { pi = __builtin_amdgcn_readlane(path[c], jl); if (lane == jl) r4c[c] = pi; }
            const int prev = col4row[pi];
            wave_lds_fence();
            if (lane == 0) col4row[pi] = j;
            j = prev;
            if (pi == cur) break;
        }
        wave_lds_fence();
    }
    return ST_OK;
}

static constexpr int MATCH_THREADS = 1024;   // 16 waves build the cost block (one query per wave at a time); wave 0 solves

__global__ __launch_bounds__(MATCH_THREADS) void matcher_kernel(
    const float* __restrict__ logits,   // [L,B,Q,K]
    const float* __restrict__ boxes,    // [L,B,Q,4] cxcywh
    const float* __restrict__ tgt_box,  // [Ttot,4] cxcywh
    const float* __restrict__ pos_map,  // [Ttot,K]
    const int* __restrict__ tgt_off,    // [B+1]
    const int* __restrict__ match_off,  // [B+1] prefix sums of min(Q,T_b)
    int L, int B, int Q, int K, float w_class, float w_bbox, float w_giou,
    long long* __restrict__ src_idx,    // [L, Mtot]
    long long* __restrict__ tgt_idx,    // [L, Mtot]
    int* __restrict__ status,           // [L*B]
    float* __restrict__ cost_out,       // optional [L, B*Q, Ttot]
    int stage_bytes)                    // > 0: LDS offset of a staging area for this image's positive-map rows and boxes
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lb = blockIdx.x;
    const int l = lb / B, b = lb % B;
    const int t0 = tgt_off[b];
    const int T = tgt_off[b + 1] - t0;
    const int Ttot = tgt_off[B];
    const int Mtot = match_off[B];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;

    if (T == 0) {
        if (tid == 0) status[lb] = ST_OK;
        return;
    }
    const bool flip = T < Q;          // SciPy solves the transpose when rows > cols
    const int R = flip ? T : Q;       // rows of the oriented problem
    const int C = flip ? Q : T;       // columns

    // ---- LDS carve -------------------------------------------------------------
    double* du = reinterpret_cast<double*>(smem);            // [R]
    double* dv = du + R;                                      // [C]
    double* spc = dv + C;                                     // [C]
    int* path = reinterpret_cast<int*>(spc + C);              // [C]
    int* row4col = path + C;                                  // [C]
    int* remaining = row4col + C;                             // [C]
    int* col4row = remaining + C;                             // [R]
    int* in_sr = col4row + R;                                 // [R]
    int* in_sc = in_sr + R;                                   // [C]
    int* bad = in_sc + C;                                     // [1]
    float* cw = reinterpret_cast<float*>(bad + 1);            // [R*C] oriented cost

    // The T positive-map rows and target boxes of this image are read once per query: from global memory every (q, t) step
    // is one exposed L2 round trip (~250 of them per wave); staged in LDS when they fit.  Same values, same arithmetic.
    const float* pm_base = pos_map + (size_t)t0 * K;
    const float* tb_base = tgt_box + (size_t)t0 * 4;
    if (stage_bytes > 0) {
        float* stage = reinterpret_cast<float*>(smem + stage_bytes);
        for (int e = tid; e < T * K; e += MATCH_THREADS) stage[e] = pm_base[e];
        for (int e = tid; e < T * 4; e += MATCH_THREADS) stage[T * K + e] = tb_base[e];
        pm_base = stage;
        tb_base = stage + T * K;
    }
    if (tid == 0) *bad = 0;
    __syncthreads();

    // ---- cost block (matcher.py:63-81) --------------------------------------------
    const float* lg = logits + ((size_t)(l * B + b) * Q) * K;
    const float* bx = boxes + ((size_t)(l * B + b) * Q) * 4;
    int my_bad = 0;
    for (int q = wave; q < Q; q += MATCH_THREADS / 64) {
        // softmax over K (exp(x - max) / sum); the row is fetched once
        const float* row = lg + (size_t)q * K;
        float pr[8];  // K <= 512
#pragma unroll
        for (int m = 0; m < 8; ++m) {
            const int k = lane + 64 * m;
            pr[m] = (k < K) ? row[k] : -INFINITY;
        }
        float mx = -INFINITY;
#pragma unroll
        for (int m = 0; m < 8; ++m) mx = fmaxf(mx, pr[m]);
        mx = wave_max(mx);
        float sum = 0.f;
#pragma unroll
        for (int m = 0; m < 8; ++m) {
            const int k = lane + 64 * m;
            pr[m] = (k < K) ? expf(pr[m] - mx) : 0.f;
            sum += pr[m];
        }
        sum = wave_sum(sum);
#pragma unroll
        for (int m = 0; m < 8; ++m) pr[m] = pr[m] / sum;

        const float cx = bx[q * 4 + 0], cy = bx[q * 4 + 1], w = bx[q * 4 + 2], h = bx[q * 4 + 3];
        const float ax0 = cx - 0.5f * w, ay0 = cy - 0.5f * h, ax1 = cx + 0.5f * w, ay1 = cy + 0.5f * h;
        const float area_a = (ax1 - ax0) * (ay1 - ay0);

        // class term: one wave-wide dot product per target; then lane t finishes target t (box L1, GIoU, weighted sum) so the
        // scalar tail runs once per query instead of once per (query, target)
        for (int tb0 = 0; tb0 < T; tb0 += 64) {
            const int tn = min(64, T - tb0);
            float mydot = 0.f;
            for (int tt = 0; tt < tn; ++tt) {
                const float* pm = pm_base + (size_t)(tb0 + tt) * K;
                float dot = 0.f;
#pragma unroll
                for (int m = 0; m < 8; ++m) {
                    const int k = lane + 64 * m;
                    if (k < K) dot += pr[m] * pm[k];
                }
                dot = wave_sum(dot);
                if (lane == tt) mydot = dot;
            }
            if (lane < tn) {
                const int t = tb0 + lane;
                const float cost_class = -mydot;
                const float* tb = tb_base + (size_t)t * 4;
                const float tcx = tb[0], tcy = tb[1], tw = tb[2], th = tb[3];
                const float cost_bbox = ((fabsf(cx - tcx) + fabsf(cy - tcy)) + fabsf(w - tw)) + fabsf(h - th);
                const float bx0 = tcx - 0.5f * tw, by0 = tcy - 0.5f * th, bx1 = tcx + 0.5f * tw, by1 = tcy + 0.5f * th;
                const float area_b = (bx1 - bx0) * (by1 - by0);
                const float iw = fmaxf(fminf(ax1, bx1) - fmaxf(ax0, bx0), 0.f);
                const float ih = fmaxf(fminf(ay1, by1) - fmaxf(ay0, by0), 0.f);
                const float inter = iw * ih;
                const float uni = (area_a + area_b) - inter;
                const float iou = inter / uni;
                const float ew = fmaxf(fmaxf(ax1, bx1) - fminf(ax0, bx0), 0.f);
                const float eh = fmaxf(fmaxf(ay1, by1) - fminf(ay0, by0), 0.f);
                const float earea = ew * eh;
                const float giou = iou - (earea - uni) / earea;
                const float cost_giou = -giou;
                const float c = (w_bbox * cost_bbox + w_class * cost_class) + w_giou * cost_giou;
                if (c != c || c == -INFINITY) my_bad = 1;
                cw[flip ? (t * Q + q) : (q * T + t)] = c;
                if (cost_out) cost_out[((size_t)l * B * Q + (size_t)b * Q + q) * Ttot + t0 + t] = c;
            }
        }
    }
    if (my_bad) atomicOr(bad, 1);
    __syncthreads();
    if (*bad) {
        if (tid == 0) status[lb] = ST_INVALID;
        return;
    }
    if (wave != 0) return;

    // (round 6) the register-resident solver for up to 256 columns: Q = 100 queries x T <= Q targets is solved on the transpose, C = Q (82 VGPRs, no spills)
    int st;
    if (C <= 128) st = lsap_solve_wave_reg<2>(cw, R, C, du, spc, col4row, in_sr, lane);
    else if (C <= 256) st = lsap_solve_wave_reg<4>(cw, R, C, du, spc, col4row, in_sr, lane);
    else st = lsap_solve_wave(cw, R, C, du, dv, spc, path, row4col, remaining, col4row, in_sr, in_sc, lane);

    if (lane == 0) status[lb] = st;
    if (st != ST_OK) return;

    long long* so = src_idx + (size_t)l * Mtot + match_off[b];
    long long* to = tgt_idx + (size_t)l * Mtot + match_off[b];
    if (flip) {
        // rows = targets; result ordered by query index (argsort of col4row)
        for (int t = lane; t < R; t += 64) {
            const int q = col4row[t];
            int rank = 0;
            for (int o = 0; o < R; ++o) rank += (col4row[o] < q);
            so[rank] = q;
            to[rank] = t;
        }
    } else {
        for (int q = lane; q < R; q += 64) { so[q] = q; to[q] = col4row[q]; }
    }
}

// Batched LSAP on caller-supplied cost matrices (scipy.optimize.linear_sum_assignment semantics: rows sorted
// ascending in the output, the transpose is solved when rows > cols, NaN / -inf entries are invalid).  One workgroup
// (one solving wavefront) per problem; problem p is [rows[p], cols[p]] fp32, row-major, at cost + offset[p].
__global__ __launch_bounds__(64) void lsap_kernel(const float* __restrict__ cost, const long long* __restrict__ offset,
                                                  const int* __restrict__ rows, const int* __restrict__ cols, const int ld,
                                                  const long long* __restrict__ out_off, long long* __restrict__ row_idx,
                                                  long long* __restrict__ col_idx, int* __restrict__ status) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int p = blockIdx.x, lane = threadIdx.x;
    const int nr = rows[p], nc = cols[p];
    if (nr == 0 || nc == 0) {
        if (lane == 0) status[p] = ST_OK;
        return;
    }
    const bool flip = nc < nr;
    const int R = flip ? nc : nr, C = flip ? nr : nc;
    double* du = reinterpret_cast<double*>(smem);
    double* dv = du + R;
    double* spc = dv + C;
    int* path = reinterpret_cast<int*>(spc + C);
    int* row4col = path + C;
    int* remaining = row4col + C;
    int* col4row = remaining + C;
    int* in_sr = col4row + R;
    int* in_sc = in_sr + R;
    float* cw = reinterpret_cast<float*>(in_sc + C);
    const float* src = cost + offset[p];
    int bad = 0;
    const int pitch = ld > 0 ? ld : nc;     // row stride of the caller's matrix (a block of a larger one when ld > cols)
    for (int e = lane; e < nr * nc; e += 64) {
        const int i = e / nc, j = e - i * nc;
        const float c = src[(long long)i * pitch + j];
        if (c != c || c == -INFINITY) bad = 1;
        cw[flip ? (j * nr + i) : e] = c;
    }
    bad = __any(bad);
    wave_lds_fence();
    if (bad) {
        if (lane == 0) status[p] = ST_INVALID;
        return;
    }
    int st;
    if (C <= 128) st = lsap_solve_wave_reg<2>(cw, R, C, du, spc, col4row, in_sr, lane);
    else if (C <= 256) st = lsap_solve_wave_reg<4>(cw, R, C, du, spc, col4row, in_sr, lane);
    else if (C <= 512) st = lsap_solve_wave_reg<8>(cw, R, C, du, spc, col4row, in_sr, lane);
    else if (C <= 1024) st = lsap_solve_wave_reg<16>(cw, R, C, du, spc, col4row, in_sr, lane);
    else st = lsap_solve_wave(cw, R, C, du, dv, spc, path, row4col, remaining, col4row, in_sr, in_sc, lane);
    if (lane == 0) status[p] = st;
    if (st != ST_OK) return;
    long long* ro = row_idx + out_off[p];
    long long* co = col_idx + out_off[p];
    if (flip) {   // solved on the transpose: pairs (row = col4row[t], col = t), sorted by row
        for (int t = lane; t < R; t += 64) {
            const int r = col4row[t];
            int rank = 0;
            for (int o = 0; o < R; ++o) rank += (col4row[o] < r);
            ro[rank] = r;
            co[rank] = t;
        }
    } else {
        for (int r = lane; r < R; r += 64) { ro[r] = r; co[r] = col4row[r]; }
    }
}

static size_t lsap_lds_bytes(int R, int C, long long cells) {
    size_t bytes = sizeof(double) * ((size_t)R + 2 * (size_t)C) + sizeof(int) * (4 * (size_t)C + 2 * (size_t)R) + sizeof(float) * (size_t)cells;
    return (bytes + 15) & ~(size_t)15;
}

static size_t matcher_lds_bytes(int Q, int maxT) {
    const int R = maxT < Q ? maxT : Q;
    const int C = maxT < Q ? Q : maxT;
    // R,C of any image are bounded by (min(Q,maxT), max(Q,maxT)); size for the worst image
    const size_t Rm = (size_t)(Q < maxT ? Q : maxT), Cm = (size_t)(Q > maxT ? Q : maxT);
    (void)R; (void)C;
    size_t bytes = sizeof(double) * (Rm + 2 * Cm) + sizeof(int) * (4 * Cm + 2 * Rm + 2) + sizeof(float) * (size_t)Q * (size_t)maxT;
    return (bytes + 15) & ~(size_t)15;
}

}  // namespace toist

extern "C" int toist_matcher(const float* logits, const float* boxes, const float* tgt_boxes, const float* pos_map,
                             const int32_t* tgt_off_dev, const int32_t* match_off_dev, int L, int B, int Q, int K,
                             int max_T, float w_class, float w_bbox, float w_giou, int64_t* src_idx,
                             int64_t* tgt_idx, int32_t* status, float* cost_out, void* stream) {
    using namespace toist;
    TOIST_REQUIRE(L > 0 && B > 0 && Q > 0 && K > 0 && K <= 512, "toist_matcher: bad shape L=%d B=%d Q=%d K=%d (K<=512)", L, B, Q, K);
    TOIST_REQUIRE(max_T >= 0, "toist_matcher: max_T < 0");
    TOIST_REQUIRE(w_class != 0.f || w_bbox != 0.f || w_giou != 0.f, "toist_matcher: all costs cant be 0");
    if (max_T == 0) {
        hipError_t e = hipMemsetAsync(status, 0, sizeof(int32_t) * (size_t)L * B, (hipStream_t)stream);
        if (e != hipSuccess) { set_last_error("toist_matcher: memset: %s", hipGetErrorString(e)); return TOIST_EHIP; }
        return TOIST_OK;
    }
    size_t lds = matcher_lds_bytes(Q, max_T);
    TOIST_REQUIRE(lds <= 160 * 1024, "toist_matcher: Q=%d max_T=%d needs %zu B of LDS (> 160 KiB)", Q, max_T, lds);
    const size_t stage = sizeof(float) * (size_t)max_T * ((size_t)K + 4);
    int stage_off = 0;
    if (lds + stage <= 64 * 1024) {   // staging is an optimisation: only while the block stays within the default LDS window
        stage_off = (int)lds;
        lds += stage;
    }
    if (lds > 64 * 1024) {   // one-time, thread-safe opt-in to the full 160 KiB LDS window (no other global state in this library)
        static std::atomic<unsigned long long> done{0};     // one bit per device: the attribute is per device
        if (!lds_attr_once_flag(done, [] { return hipFuncSetAttribute((const void*)matcher_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) == hipSuccess; })) {
            set_last_error("toist_matcher: cannot enable 160 KiB of LDS");
            return TOIST_EHIP;
        }
    }
    hipLaunchKernelGGL(matcher_kernel, dim3(L * B), dim3(MATCH_THREADS), lds, (hipStream_t)stream, logits, boxes, tgt_boxes,
                       pos_map, tgt_off_dev, match_off_dev, L, B, Q, K, w_class, w_bbox, w_giou,
                       (long long*)src_idx, (long long*)tgt_idx, status, cost_out, stage_off);
    return check_launch("toist_matcher");
}

extern "C" int toist_lsap(const float* cost, const int64_t* offset, const int32_t* rows, const int32_t* cols, int ld, int n, int max_rows,
                          int max_cols, int64_t max_cells, const int64_t* out_off, int64_t* row_idx, int64_t* col_idx, int32_t* status,
                          void* stream) {
    using namespace toist;
    TOIST_REQUIRE(cost && offset && rows && cols && out_off && row_idx && col_idx && status && n > 0, "toist_lsap: bad args");
    TOIST_REQUIRE(max_rows >= 0 && max_cols >= 0 && max_cells >= 0 && (ld == 0 || ld >= max_cols), "toist_lsap: negative extent or ld < cols");
    const int R = max_rows < max_cols ? max_rows : max_cols, C = max_rows < max_cols ? max_cols : max_rows;
    const size_t lds = lsap_lds_bytes(R, C, max_cells);
    TOIST_REQUIRE(lds <= 160 * 1024, "toist_lsap: problems up to %d x %d (%lld cells) need %zu B of LDS (> 160 KiB)", max_rows, max_cols,
                  (long long)max_cells, lds);
    if (lds > 64 * 1024) {
        static std::atomic<unsigned long long> done{0};
        if (!lds_attr_once_flag(done, [] { return hipFuncSetAttribute((const void*)lsap_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) == hipSuccess; })) {
            set_last_error("toist_lsap: cannot enable 160 KiB of LDS");
            return TOIST_EHIP;
        }
    }
    hipLaunchKernelGGL(lsap_kernel, dim3(n), dim3(64), lds, (hipStream_t)stream, cost, (const long long*)offset, rows, cols, ld,
                       (const long long*)out_off, (long long*)row_idx, (long long*)col_idx, status);
    return check_launch("toist_lsap");
}
