// Deterministic batch-1 mode: ONE wavefront consumes the tuples strictly in order.
//
// Lane l owns the latent components k = l, l+64, ... of every row, so each statement of the
// reference's updateFactors becomes one load/compute/store by the lane that owns k, issued in the
// reference's statement order directly on device memory.  Rows that alias (prev == pos, a repeated
// history item, ...) therefore behave exactly as the reference's in-place updates do: a lane always
// re-reads what it last wrote.  Only the K-long reductions cross lanes (wave shuffles), and scalars
// (biases, WT, alpha) are owned by lane 0 and broadcast.
#include "common.h"

#define E REC_MAX_EPL

template <class T>
struct Seq {
    const DevModel& m;
    int lane;
    T lr;
    __device__ Seq(const DevModel& m_, T lr_) : m(m_), lane(threadIdx.x & 63), lr(lr_) {}
    __device__ T* tab(int t) const { return (T*)m.tab[t]; }
    __device__ static T wsum(T v) { return group_sum<64>(v); }
    __device__ static T bcast0(T v) { return __shfl(v, 0, 64); }
    // value of a scalar owned by lane 0
    __device__ T scalar(const T* p) const { T v = 0; if (lane == 0) v = *p; return bcast0(v); }
    __device__ T dot(const T* a, const T* b, int K) const
    {
        T s = 0;
        for (int k = lane; k < K; k += 64) s += a[k] * b[k];
        return wsum(s);
    }

    // BPRMF.cpp:85-105
    __device__ void bprmf(int u, int pos, int neg)
    {
        const int K = m.K; const T lam = (T)m.lambda, breg = (T)m.bias_reg;
        T* beta = tab(BPRMF_BETA); T* gu = tab(BPRMF_GU) + (int64_t)u * K;
        T* gp = tab(BPRMF_GI) + (int64_t)pos * K; T* gn = tab(BPRMF_GI) + (int64_t)neg * K;
        T x = scalar(beta + pos) - scalar(beta + neg);
        x += dot(gu, gp, K) - dot(gu, gn, K);
        T deri = 1 / (1 + rexp(x));
        if (lane == 0) {
            beta[pos] += lr * (deri - breg * beta[pos]);
            beta[neg] += lr * (-deri - breg * beta[neg]);
        }
        for (int f = lane; f < K; f += 64) {
            T w_uf = gu[f], h_if = gp[f], h_jf = gn[f];
            gu[f] += lr * (deri * (h_if - h_jf) - lam * w_uf);
            gp[f] += lr * (deri * w_uf - lam * h_if);
            gn[f] += lr * (-deri * w_uf - lam / (T)10.0 * h_jf);
        }
    }

    // MC.cpp:91-110
    __device__ void mc(int i, int pos, int neg)
    {
        const int K = m.K; const T lam = (T)m.lambda;
        T* g = tab(MC_GI) + (int64_t)i * K;
        T* ep = tab(MC_ETA) + (int64_t)pos * K; T* en = tab(MC_ETA) + (int64_t)neg * K;
        T pmn[E]; T s = 0;
#pragma unroll
        for (int j = 0; j < E; j++) { int k = lane + 64 * j; pmn[j] = 0; if (k < K) { pmn[j] = ep[k] - en[k]; s += g[k] * pmn[j]; } }
        T deri = (T)1.0 / ((T)1.0 + rexp(wsum(s)));
#pragma unroll
        for (int j = 0; j < E; j++) {
            int f = lane + 64 * j;
            if (f < K) {
                T deri_A = deri * g[f];
                g[f] += lr * (deri * pmn[j] - lam * g[f]);
                ep[f] += lr * (deri_A - lam * ep[f]);
                en[f] += lr * (-deri_A - lam * en[f]);
            }
        }
    }

    // FPMC.cpp:88-122
    __device__ void fpmc(int u, int prev, int pos, int neg)
    {
        const int K = m.K, KK = m.KK; const T lam = (T)m.lambda;
        T* gu = tab(FPMC_GU) + (int64_t)u * K;
        T* gp = tab(FPMC_GI) + (int64_t)pos * K; T* gn = tab(FPMC_GI) + (int64_t)neg * K;
        T* ka = tab(FPMC_KAPPA) + (int64_t)prev * KK;
        T* ep = tab(FPMC_ETA) + (int64_t)pos * KK; T* en = tab(FPMC_ETA) + (int64_t)neg * KK;
        T a[E], b[E]; T s1 = 0, s2 = 0;
#pragma unroll
        for (int j = 0; j < E; j++) {
            int k = lane + 64 * j; a[j] = b[j] = 0;
            if (k < K) { a[j] = gp[k] - gn[k]; s1 += gu[k] * a[j]; }
            if (k < KK) { b[j] = ep[k] - en[k]; s2 += ka[k] * b[j]; }
        }
        T deri = (T)1.0 / ((T)1.0 + rexp(wsum(s1) + wsum(s2)));
#pragma unroll
        for (int j = 0; j < E; j++) {
            int f = lane + 64 * j;
            if (f < K) {
                T deri_A = deri * gu[f];
                gu[f] += lr * (deri * a[j] - lam * gu[f]);
                gp[f] += lr * (deri_A - lam * gp[f]);
                gn[f] += lr * (-deri_A - lam * gn[f]);
            }
        }
#pragma unroll
        for (int j = 0; j < E; j++) {
            int f = lane + 64 * j;
            if (f < KK) {
                T deri_B = deri * ka[f];
                ka[f] += lr * (deri * b[j] - lam * ka[f]);
                ep[f] += lr * (deri_B - lam * ep[f]);
                en[f] += lr * (-deri_B - lam * en[f]);
            }
        }
    }

    __device__ T l2sq(const T* a, const T* b, int K) const
    {
        T s = 0;
        for (int k = lane; k < K; k += 64) { T d = a[k] - b[k]; s += d * d; }
        return wsum(s);
    }

    // PRME.cpp:82-103 (prediction: PRME.cpp:64-69)
    __device__ void prme(int u, int prev, int pos, int neg)
    {
        const int K = m.K, KK = m.KK; const T lam = (T)m.lambda, alpha = (T)m.alpha;
        T* gu = tab(PRME_GU) + (int64_t)u * K;
        T* gp = tab(PRME_GI) + (int64_t)pos * K; T* gn = tab(PRME_GI) + (int64_t)neg * K;
        T* ei = tab(PRME_ETA) + (int64_t)prev * KK;
        T* ep = tab(PRME_ETA) + (int64_t)pos * KK; T* en = tab(PRME_ETA) + (int64_t)neg * KK;
        T xp = -(alpha * l2sq(gu, gp, K) + (1 - alpha) * l2sq(ei, ep, KK));
        T xn = -(alpha * l2sq(gu, gn, K) + (1 - alpha) * l2sq(ei, en, KK));
        T deri = (T)1.0 / ((T)1.0 + rexp(xp - xn));
        for (int f = lane; f < K; f += 64) {
            T g_uf = gu[f];
            gu[f] += lr * (deri * alpha * (gp[f] - gn[f]) - lam * g_uf);
            gp[f] += lr * (deri * alpha * (g_uf - gp[f]) - lam * gp[f]);
            gn[f] += lr * (deri * (-alpha) * (g_uf - gn[f]) - lam * gn[f]);
        }
        for (int f = lane; f < KK; f += 64) {
            T e_if = ei[f];
            ei[f] += lr * (deri * (1 - alpha) * (ep[f] - en[f]) - lam * e_if);
            ep[f] += lr * (deri * (1 - alpha) * (e_if - ep[f]) - lam * ep[f]);
            en[f] += lr * (deri * (alpha - 1) * (e_if - en[f]) - lam * en[f]);
        }
    }

    // HRM_avg.cpp:168-195, HRM_max.cpp:178-208 (the negative is part of the tuple)
    __device__ void hrm(int u, int prev, int item, int neg, bool is_max)
    {
        const int K = m.K; const T lam = (T)m.lambda;
        T* Uu = tab(HRM_U) + (int64_t)u * K; T* Vp = tab(HRM_V) + (int64_t)prev * K;
        T* Vi = tab(HRM_V) + (int64_t)item * K; T* Vn = tab(HRM_V) + (int64_t)neg * K;
        T h[E]; T s1 = 0, s2 = 0;
#pragma unroll
        for (int j = 0; j < E; j++) {
            int k = lane + 64 * j; h[j] = 0;
            if (k < K) {
                T a = Uu[k], b = Vp[k];
                h[j] = is_max ? (a > b ? a : b) : (a + b);
                s1 += h[j] * Vi[k]; s2 += h[j] * Vn[k];
            }
        }
        T deri = (T)1.0 / ((T)1.0 + rexp(wsum(s1) - wsum(s2)));
#pragma unroll
        for (int j = 0; j < E; j++) {
            int k = lane + 64 * j;
            if (k < K) {
                T deri_k = deri * (Vi[k] - Vn[k]);
                Vi[k] += lr * (deri * h[j] - lam * Vi[k]);
                Vn[k] += lr * (-deri * h[j] - lam * Vn[k]);
                if (!is_max) {
                    Uu[k] += lr * (deri_k - lam * Uu[k]);
                    Vp[k] += lr * (deri_k - lam * Vp[k]);
                } else if (Uu[k] > Vp[k]) {
                    Uu[k] += lr * (deri_k - lam * Uu[k]);
                } else {
                    Vp[k] += lr * (deri_k - lam * Vp[k]);
                }
            }
        }
    }

    // TransRec.cpp:245-258
    __device__ void unit_ball(T* row, int K)
    {
        T s = 0;
        for (int k = lane; k < K; k += 64) s += row[k] * row[k];
        s = rsqrt_(wsum(s));
        if (s > 1) for (int k = lane; k < K; k += 64) row[k] /= s;
    }

    // TransRec.cpp:208-243, TransRec_L1.cpp:207-244
    __device__ void transrec(int u, int x, int y, int yn, bool l1)
    {
        const int K = m.K; const T lam = (T)m.lambda, breg = (T)m.bias_reg, rreg = (T)m.relation_reg;
        T* beta = tab(TR_BETA); T* r = tab(TR_r); T* R = tab(TR_R) + (int64_t)u * K;
        T* Hx = tab(TR_H) + (int64_t)x * K; T* Hy = tab(TR_H) + (int64_t)y * K; T* Hn = tab(TR_H) + (int64_t)yn * K;
        T v[E], vn[E]; T zs = 0;
#pragma unroll
        for (int j = 0; j < E; j++) {
            int k = lane + 64 * j; v[j] = vn[j] = 0;
            if (k < K) {
                v[j] = r[k] + R[k] + Hx[k] - Hy[k];
                vn[j] = r[k] + R[k] + Hx[k] - Hn[k];
                zs += l1 ? (rabs(v[j]) - rabs(vn[j])) : (v[j] * v[j] - vn[j] * vn[j]);
            }
        }
        T z = -scalar(beta + y) + scalar(beta + yn);
        z -= wsum(zs);
        T deri = 1 / (1 + rexp(z));
        if (lane == 0) {
            beta[y] += lr * (-deri - breg * beta[y]);
            beta[yn] += lr * (deri - breg * beta[yn]);
        }
#pragma unroll
        for (int j = 0; j < E; j++) {
            int k = lane + 64 * j;
            if (k < K) {
                if (!l1) {
                    Hx[k] += lr * (deri * 2 * (-v[j] + vn[j]) - lam * Hx[k]);
                    Hn[k] += lr * (deri * 2 * (-vn[j]) - lam * Hn[k]);
                    Hy[k] += lr * (deri * 2 * (v[j]) - lam * Hy[k]);
                    r[k] += lr * (deri * 2 * (-v[j] + vn[j]) - lam * r[k]);
                    R[k] += lr * (deri * 2 * (-v[j] + vn[j]) - rreg * R[k]);
                } else {
                    T tmp = (T)((v[j] > 0 ? -1 : 1) + (vn[j] > 0 ? 1 : -1));
                    Hx[k] += lr * deri * tmp;
                    Hy[k] += lr * deri * (T)(v[j] > 0 ? 1 : -1);
                    Hn[k] += lr * deri * (T)(vn[j] > 0 ? -1 : 1);
                    r[k] += lr * (deri * tmp - lam * r[k]);
                    R[k] += lr * (deri * tmp - rreg * R[k]);
                }
            }
        }
        unit_ball(Hx, K); unit_ball(Hy, K); unit_ball(Hn, K);
    }

    // FossilSimple.cpp:151-235
    __device__ void fossil_simple(int u, int prev, int pos, int neg)
    {
        const int K = m.K; const T lam = (T)m.lambda, breg = (T)m.bias_reg;
        T* alpha = tab(FS_ALPHA); T* alpha_u = tab(FS_ALPHA_U); T* beta = tab(FS_BETA);
        T* Ui = tab(FS_UI); T* V = tab(FS_V);
        T* Vp = V + (int64_t)pos * K; T* Vn = V + (int64_t)neg * K; T* Up = Ui + (int64_t)prev * K;
        const int32_t* h = m.seq_item + m.seq_off[u]; const int n = (int)(m.seq_off[u + 1] - m.seq_off[u]);
        T sp[E], sn[E];
        double spd[E], snd[E];   // sums over a whole history: fp64 whatever the tables' type (thousands of f32 additions lose ~1e-4 of the sum)
#pragma unroll
        for (int j = 0; j < E; j++) { sp[j] = sn[j] = 0; spd[j] = snd[j] = 0; }
        int cp = 0, cn = 0;
        for (int t = 0; t < n; t++) {
            const int it = h[t]; const T* Uh = Ui + (int64_t)it * K;
            if (it != pos) cp++;
            if (it != neg) cn++;
#pragma unroll
            for (int j = 0; j < E; j++) {
                int k = lane + 64 * j;
                if (k < K) { const double x = (double)Uh[k]; if (it != pos) spd[j] += x; if (it != neg) snd[j] += x; }
            }
        }
        T wp = cp > 0 ? rpow((T)cp, (T)-0.2) : 0, wn = cn > 0 ? rpow((T)cn, (T)-0.2) : 0;
        T a0 = scalar(alpha) + scalar(alpha_u + u);
        T d1 = 0, d2 = 0;
#pragma unroll
        for (int j = 0; j < E; j++) {
            int k = lane + 64 * j;
            if (k < K) {
                sp[j] = (T)((double)wp * spd[j]) + a0 * Up[k];
                sn[j] = (T)((double)wn * snd[j]) + a0 * Up[k];
                d1 += sp[j] * Vp[k]; d2 += sn[j] * Vn[k];
            }
        }
        T x = scalar(beta + pos) - scalar(beta + neg);
        x += wsum(d1) - wsum(d2);
        T deri = 1 / (1 + rexp(x));
        T g = dot(Up, Vp, K) - dot(Up, Vn, K);
        if (lane == 0) {
            beta[pos] += lr * (deri - breg * beta[pos]);
            beta[neg] += lr * (-deri - breg * beta[neg]);
            *alpha += lr / 10 * (deri * g - lam / 10 * (*alpha));
            alpha_u[u] += lr * (deri * g - lam * alpha_u[u]);
        }
        const T prev_alpha = a0;
        for (int t = 0; t < n; t++) {
            const int it = h[t]; T* Uh = Ui + (int64_t)it * K;
            for (int k = lane; k < K; k += 64) {
                if (it == prev) Uh[k] += lr * (deri * ((wp + prev_alpha) * Vp[k] - (wn + prev_alpha) * Vn[k]) - lam * Uh[k]);
                else if (it != pos) {
                    if (it != neg) Uh[k] += lr * (deri * (wp * Vp[k] - wn * Vn[k]) - lam * Uh[k]);
                    else Uh[k] += lr * (deri * wp * Vp[k] - lam * Uh[k]);
                } else Uh[k] += lr * (-deri * wn * Vn[k] - lam * Uh[k]);
            }
        }
#pragma unroll
        for (int j = 0; j < E; j++) {
            int k = lane + 64 * j;
            if (k < K) {
                Vp[k] += lr * (deri * sp[j] - lam * Vp[k]);
                Vn[k] += lr * (-deri * sn[j] - lam * Vn[k]);
            }
        }
    }

    // Fossil.cpp:130-256; prev_items of Fossil.cpp:266-272 (newest first, at most L)
    __device__ void fossil(int u, int tpos, int neg)
    {
        const int K = m.K, L = m.L; const T lam = (T)m.lambda, breg = (T)m.bias_reg, df = (T)m.df;
        T* beta = tab(FO_BETA); T* WT = tab(FO_WT); T* WTu = tab(FO_WTU) + (int64_t)u * L;
        T* Ui = tab(FO_UI); T* V = tab(FO_V);
        const int32_t* h = m.seq_item + m.seq_off[u]; const int n = (int)(m.seq_off[u + 1] - m.seq_off[u]);
        const int pos = h[tpos]; const int np = tpos < L ? tpos : L;
        T* Vp = V + (int64_t)pos * K; T* Vn = V + (int64_t)neg * K;
        T sp[E], sn[E];
        double spd[E], snd[E];   // sums over a whole history: fp64 whatever the tables' type (thousands of f32 additions lose ~1e-4 of the sum)
#pragma unroll
        for (int j = 0; j < E; j++) { sp[j] = sn[j] = 0; spd[j] = snd[j] = 0; }
        int cp = 0, cn = 0;
        for (int t = 0; t < n; t++) {
            const int it = h[t]; const T* Uh = Ui + (int64_t)it * K;
            if (it != pos) cp++;
            if (it != neg) cn++;
#pragma unroll
            for (int j = 0; j < E; j++) {
                int k = lane + 64 * j;
                if (k < K) { const double x = (double)Uh[k]; if (it != pos) spd[j] += x; if (it != neg) snd[j] += x; }
            }
        }
        T wp = cp > 0 ? rpow((T)cp, (T)-0.2) : 0, wn = cn > 0 ? rpow((T)cn, (T)-0.2) : 0;
#pragma unroll
        for (int j = 0; j < E; j++) { sp[j] = (T)((double)wp * spd[j]); sn[j] = (T)((double)wn * snd[j]); }
        for (int ind = 0; ind < np; ind++) {
            const int c = h[tpos - 1 - ind]; const T* Uc = Ui + (int64_t)c * K;
            const T w = scalar(WT + ind) + scalar(WTu + ind);
#pragma unroll
            for (int j = 0; j < E; j++) {
                int k = lane + 64 * j;
                if (k < K) { if (c != pos) sp[j] += w * Uc[k]; if (c != neg) sn[j] += w * Uc[k]; }
            }
        }
        T d1 = 0, d2 = 0;
#pragma unroll
        for (int j = 0; j < E; j++) { int k = lane + 64 * j; if (k < K) { d1 += sp[j] * Vp[k]; d2 += sn[j] * Vn[k]; } }
        T x = scalar(beta + pos) - scalar(beta + neg);
        x += wsum(d1) - wsum(d2);
        T deri = 1 / (1 + rexp(x));
        if (lane == 0) {
            beta[pos] += lr * (deri - breg * beta[pos]);
            beta[neg] += lr * (-deri - breg * beta[neg]);
        }
        T wts[REC_MAX_L];
        for (int ind = 0; ind < np; ind++) {
            wts[ind] = scalar(WT + ind) + scalar(WTu + ind);
            const int c = h[tpos - 1 - ind]; const T* Uc = Ui + (int64_t)c * K;
            const T dvp = dot(Uc, Vp, K), dvn = dot(Uc, Vn, K);
            if (lane == 0) {
                if (c != pos) {
                    if (c != neg) {
                        T diff = dvp - dvn;
                        WT[ind] += lr / df * (deri * diff - lam / df * WT[ind]);
                        WTu[ind] += lr * (deri * diff - lam * WTu[ind]);
                    } else {
                        WT[ind] += lr / df * (deri * dvp - lam / df * WT[ind]);
                        WTu[ind] += lr * (deri * dvp - lam * WT[ind]);   // WT, not WTu: Fossil.cpp:201
                    }
                } else {
                    WT[ind] += lr / df * (-deri * dvn - lam / df * WT[ind]);
                    WTu[ind] += lr * (-deri * dvn - lam * WT[ind]);      // Fossil.cpp:205
                }
            }
        }
        for (int t = 0; t < n; t++) {
            const int it = h[t]; T* Uh = Ui + (int64_t)it * K;
            int pt = -1;
            for (int ind = 0; ind < np; ind++) if (h[tpos - 1 - ind] == it) { pt = ind; break; }
            const T a = pt >= 0 ? wts[pt] : (T)0;
            for (int k = lane; k < K; k += 64) {
                if (pt >= 0) {
                    if (it != pos) {
                        if (it != neg) Uh[k] += lr * (deri * ((wp + a) * Vp[k] - (wn + a) * Vn[k]) - lam * Uh[k]);
                        else Uh[k] += lr * (deri * (wp + a) * Vp[k] - lam * Uh[k]);
                    } else Uh[k] += lr * (-deri * (wn + a) * Vn[k] - lam * Uh[k]);
                } else {
                    if (it != pos) {
                        if (it != neg) Uh[k] += lr * (deri * (wp * Vp[k] - wn * Vn[k]) - lam * Uh[k]);
                        else Uh[k] += lr * (deri * wp * Vp[k] - lam * Uh[k]);
                    } else Uh[k] += lr * (-deri * wn * Vn[k] - lam * Uh[k]);
                }
            }
        }
#pragma unroll
        for (int j = 0; j < E; j++) {
            int k = lane + 64 * j;
            if (k < K) {
                Vp[k] += lr * (deri * sp[j] - lam * Vp[k]);
                Vn[k] += lr * (-deri * sn[j] - lam * Vn[k]);
            }
        }
    }

    // REBUS.cpp:282-444 with the context of REBUS.cpp:540-544.  The prefix SET is the prefix
    // positions flagged as first occurrences.
    __device__ void rebus(int u, int tpos, int neg)
    {
        const int K = m.K, I = m.I; const T lam = (T)m.lambda, breg = (T)m.bias_reg;
        T* beta = tab(RB_BETA); T* P = tab(RB_P);
        const int64_t so = m.seq_off[u];
        const int32_t* seq = m.seq_item + so; const uint8_t* fo = m.first_occ + so;
        const int pos = seq[tpos];
        const int64_t ctx = so + 2 * (int64_t)u + (tpos - 1);
        const int32_t* path = m.path_item + m.path_off[ctx]; const int np = (int)(m.path_off[ctx + 1] - m.path_off[ctx]);
        const double* etas = m.eta + (int64_t)np * m.L;
        T p_pos[E], p_neg[E], sp[E], sn[E], sl[E], dp[E], dn[E];
        double spd[E], snd[E];   // sums over the whole prefix: fp64 whatever the tables' type
#pragma unroll
        for (int j = 0; j < E; j++) {
            int k = lane + 64 * j; sp[j] = sn[j] = sl[j] = p_pos[j] = p_neg[j] = 0; spd[j] = snd[j] = 0;
            if (k < K) { p_pos[j] = P[(int64_t)pos * K + k]; p_neg[j] = P[(int64_t)neg * K + k]; }
        }
        int cp = 0, cn = 0;
        for (int t = 0; t < tpos; t++) {
            if (!fo[t]) continue;
            const int it = seq[t]; const T* Ph = P + (int64_t)it * K;
            if (it != pos) cp++;
            if (it != neg) cn++;
#pragma unroll
            for (int j = 0; j < E; j++) {
                int k = lane + 64 * j;
                if (k < K) { const double x = (double)Ph[k]; if (it != pos) spd[j] += x; if (it != neg) snd[j] += x; }
            }
        }
        T wp = cp > 0 ? rpow((T)cp, (T)m.alpha) : 0, wn = cn > 0 ? rpow((T)cn, (T)m.alpha) : 0;
        for (int r = 0; r < np; r++) {
            const T* Pr = P + (int64_t)path[r] * K; const T e = (T)etas[r];
#pragma unroll
            for (int j = 0; j < E; j++) { int k = lane + 64 * j; if (k < K) sl[j] += e * Pr[k]; }
        }
        // the squared lengths of the two distance vectors grow with the history and x is their difference: fp64 throughout
        const double wpd = cp > 0 ? pow((double)cp, (double)m.alpha) : 0.0, wnd = cn > 0 ? pow((double)cn, (double)m.alpha) : 0.0;
        double qd = 0;
#pragma unroll
        for (int j = 0; j < E; j++) {
            int k = lane + 64 * j; dp[j] = dn[j] = 0;
            if (k < K) {
                const double a = wpd * spd[j] + (double)sl[j] - (double)p_pos[j], b = wnd * snd[j] + (double)sl[j] - (double)p_neg[j];
                dp[j] = (T)a; dn[j] = (T)b;
                qd += b * b - a * a;
            }
        }
        T x = (-scalar(beta + pos) + scalar(beta + neg)) + (T)group_sum<64>(qd);
        T deri = 1 / (1 + rexp(x));
        if (lane == 0) {
            beta[pos] += lr * (-deri - breg * beta[pos]);   // [9]
            beta[neg] += lr * (deri - breg * beta[neg]);    // [10]
        }
        bool pos_in = false, neg_in = false;
        for (int t = 0; t < tpos; t++) {
            if (!fo[t]) continue;
            const int it = seq[t]; T* Ph = P + (int64_t)it * K;
            if (it == pos) pos_in = true;
            if (it == neg) neg_in = true;
            bool inpath = false; T se = 0;
            for (int r = 0; r < np; r++) if (path[r] == it) { inpath = true; se += (T)etas[r]; }
#pragma unroll
            for (int j = 0; j < E; j++) {
                int k = lane + 64 * j;
                if (k >= K) continue;
                if (inpath) {
                    if (it != pos) {
                        if (it != neg) Ph[k] += lr * (deri * (-(wp + se) * dp[j] + (wn + se) * dn[j]) - lam * Ph[k]);      // [1]
                    } else Ph[k] += lr * (deri * ((1 - se) * dp[j] + (wn + se) * dn[j]) - lam * p_pos[j]);                 // [4]
                } else {
                    if (it != pos) {
                        if (it != neg) Ph[k] += lr * (deri * (-wp * dp[j] + wn * dn[j]) - lam * Ph[k]);                    // [2]
                        else Ph[k] += lr * (deri * (-wp * dp[j] - dn[j]) - lam * p_neg[j]);                                // [7]
                    } else Ph[k] += lr * (deri * (dp[j] + wn * dn[j]) - lam * p_pos[j]);                                   // [5]
                }
            }
        }
#pragma unroll
        for (int j = 0; j < E; j++) {
            int k = lane + 64 * j;
            if (k >= K) continue;
            if (path[0] == I) P[(int64_t)I * K + k] += lr * (deri * (-dp[j] + dn[j]) - lam * P[(int64_t)I * K + k]);       // [3]
            if (!neg_in) P[(int64_t)neg * K + k] += lr * (deri * (-dn[j]) - lam * p_neg[j]);                               // [8]
            if (!pos_in) P[(int64_t)pos * K + k] += lr * (deri * (dp[j]) - lam * p_pos[j]);                                // [6]
        }
    }
};

template <class T>
__global__ void __launch_bounds__(64) sgd_sequential_kernel(DevModel m, const int4* __restrict__ tuples, int64_t n, T lr)
{
    Seq<T> s(m, lr);
    for (int64_t i = 0; i < n; i++) {
        const int4 t = tuples[i];
        switch (m.model) {
        case REC_BPRMF: s.bprmf(t.x, t.y, t.z); break;
        case REC_MC: s.mc(t.x, t.y, t.z); break;
        case REC_FPMC: s.fpmc(t.x, t.y, t.z, t.w); break;
        case REC_PRME: s.prme(t.x, t.y, t.z, t.w); break;
        case REC_HRM_AVG: s.hrm(t.x, t.y, t.z, t.w, false); break;
        case REC_HRM_MAX: s.hrm(t.x, t.y, t.z, t.w, true); break;
        case REC_TRANSREC: s.transrec(t.x, t.y, t.z, t.w, false); break;
        case REC_TRANSREC_L1: s.transrec(t.x, t.y, t.z, t.w, true); break;
        case REC_FOSSIL_SIMPLE: s.fossil_simple(t.x, t.y, t.z, t.w); break;
        case REC_FOSSIL: s.fossil(t.x, t.y, t.z); break;
        case REC_REBUS: s.rebus(t.x, t.y + 1, t.z); break;
        default: break;
        }
    }
}

int launch_sgd_sequential(const DevModel& m, int precision, const int4* tuples, int64_t n, double lr, hipStream_t st)
{
    if (n <= 0) return 0;
    if (precision == REC_F64)
        hipLaunchKernelGGL(sgd_sequential_kernel<double>, dim3(1), dim3(64), 0, st, m, tuples, n, lr);
    else
        hipLaunchKernelGGL(sgd_sequential_kernel<float>, dim3(1), dim3(64), 0, st, m, tuples, n, (float)lr);
    HIP_TRY(hipGetLastError());
    return 0;
}
