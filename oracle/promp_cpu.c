/* promp_cpu.c -- TEST / MEASUREMENT INFRASTRUCTURE ONLY: a C + OpenMP restatement of the hot path, used by bench.py's
 * `cpu_baseline` leg ("kind": "port") and pinned against oracle/*.py by tests/test_oracle_cpu_port.py.  Nothing under
 * promp_amd/ may load it.
 *
 * What it restates (paths relative to /root/reference/meta_policy_search/), one OpenMP thread per task the way the
 * reference's TF graph runs its per-task sub-graphs side by side:
 *   samplers/base.py:99-162, utils/utils.py:59-81, baselines/linear_baseline.py:55-106   returns, features, ridge fit
 *                                                                                        (float64), GAE, normalisation
 *   policies/networks/mlp.py:65-119, policies/distributions/diagonal_gaussian.py:16-109  tanh MLP, Gaussian log-lik / KL
 *   meta_algos/pro_mp.py:59-155, meta_algos/base.py:192-215                              inner step, meta-objective and its
 *                                                                                        exact gradient (R-operator HVP)
 *   meta_algos/trpo_maml.py:58-62,135                                                    log-likelihood inner / ratio / KL outer
 *   optimizers/maml_first_order_optimizer.py:22-115 (tf.train.AdamOptimizer)             Adam
 * Policy arithmetic in float32 like the TF graph; fixed-length paths (the synthetic batches of bench.py).
 *
 * Build: gcc -O3 -march=native -fopenmp -shared -fPIC oracle/promp_cpu.c -o oracle/_build/libpromp_cpu.so -lm
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

typedef struct {
    int M, P, T, O, A, H1, H2;
} pc_dims;

enum { PC_RATIO = 0, PC_CLIP = 1, PC_LOGLIK = 2, PC_KL = 3 };
#define PC_RB 64           /* rows per block */
#define PC_MAXH 128
#define PC_MAXA 8
#define PC_LOG_2PI 1.8378770664093453f

void pc_set_threads(int n) {
#ifdef _OPENMP
    omp_set_num_threads(n < 1 ? 1 : n);
#else
    (void)n;
#endif
}
int pc_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

static int n_params(const pc_dims* d) { return d->O * d->H1 + d->H1 + d->H1 * d->H2 + d->H2 + d->H2 * d->A + d->A + d->A; }

typedef struct {
    const float *W1, *b1, *W2, *b2, *W3, *b3, *s;
} net_t;
static net_t net_of(const pc_dims* d, const float* th) {
    net_t n;
    n.W1 = th; n.b1 = n.W1 + d->O * d->H1; n.W2 = n.b1 + d->H1; n.b2 = n.W2 + d->H1 * d->H2;
    n.W3 = n.b2 + d->H2; n.b3 = n.W3 + d->H2 * d->A; n.s = n.b3 + d->A;
    return n;
}

/* Z[r][j] = b[j] + sum_k X[r][k] W[k][j] */
static void affine(int R, int K, int J, const float* X, int xs, const float* W, const float* b, float* Z, int zs) {
    for (int r = 0; r < R; ++r) {
        float* z = Z + r * zs;
        for (int j = 0; j < J; ++j) z[j] = b ? b[j] : 0.f;
        for (int k = 0; k < K; ++k) {
            const float a = X[r * xs + k];
            const float* w = W + k * J;
            for (int j = 0; j < J; ++j) z[j] += a * w[j];
        }
    }
}
/* Z[r][j] += sum_k X[r][k] W[k][j] */
static void matmul_acc(int R, int K, int J, const float* X, int xs, const float* W, float* Z, int zs) {
    for (int r = 0; r < R; ++r) {
        float* z = Z + r * zs;
        for (int k = 0; k < K; ++k) {
            const float a = X[r * xs + k];
            const float* w = W + k * J;
            for (int j = 0; j < J; ++j) z[j] += a * w[j];
        }
    }
}
/* G[k][j] += sum_r X[r][k] D[r][j] ;  gb[j] += sum_r D[r][j] (gb may be NULL) */
static void outer_acc(int R, int K, int J, const float* X, int xs, const float* D, int ds, float* G, float* gb) {
    for (int r = 0; r < R; ++r) {
        const float* dd = D + r * ds;
        for (int k = 0; k < K; ++k) {
            const float a = X[r * xs + k];
            float* g = G + k * J;
            for (int j = 0; j < J; ++j) g[j] += a * dd[j];
        }
        if (gb)
            for (int j = 0; j < J; ++j) gb[j] += dd[j];
    }
}
/* Y[r][k] (+)= sum_j D[r][j] W[k][j] */
static void matmul_t(int R, int K, int J, const float* D, int ds, const float* W, float* Y, int ys, int acc) {
    for (int r = 0; r < R; ++r)
        for (int k = 0; k < K; ++k) {
            const float* w = W + k * J;
            const float* dd = D + r * ds;
            float s = 0.f;
            for (int j = 0; j < J; ++j) s += dd[j] * w[j];
            Y[r * ys + k] = acc ? Y[r * ys + k] + s : s;
        }
}

typedef struct {
    const float *obs, *act, *adv, *om, *ols; /* ols: [A] of the task */
    int N;
} slab_t;

/* loss, mean KL, gradient of the objective (grad, may be NULL) and of the mean KL (gkl, may be NULL) of ONE task */
static void loss_grad(const pc_dims* d, const float* th, const slab_t* sl, int kind, int clip_ls, float clip_eps, float* loss_out,
                      float* kl_out, float* grad, float* gkl) {
    const int O = d->O, A = d->A, H1 = d->H1, H2 = d->H2, N = sl->N, NP = n_params(d);
    const net_t n = net_of(d, th);
    const float min_ls = logf(1e-6f);
    float s[PC_MAXA], e[PC_MAXA], sn2[PC_MAXA], mask[PC_MAXA], so2[PC_MAXA], eo[PC_MAXA];
    for (int a = 0; a < A; ++a) {
        const int clipped = clip_ls && n.s[a] < min_ls;
        s[a] = clipped ? min_ls : n.s[a];
        mask[a] = clipped ? 0.f : 1.f;
        e[a] = expf(-s[a]);
        sn2[a] = expf(2.f * s[a]);
        so2[a] = expf(2.f * sl->ols[a]);
        eo[a] = expf(-sl->ols[a]);
    }
    if (grad) memset(grad, 0, sizeof(float) * NP);
    if (gkl) memset(gkl, 0, sizeof(float) * NP);
    float h1[PC_RB * PC_MAXH], h2[PC_RB * PC_MAXH], mu[PC_RB * PC_MAXA], dmu[PC_RB * PC_MAXA], dk[PC_RB * PC_MAXA];
    float dz2[PC_RB * PC_MAXH], dz1[PC_RB * PC_MAXH];
    double loss = 0.0, kls = 0.0;
    const float invN = 1.f / (float)N;
    for (int r0 = 0; r0 < N; r0 += PC_RB) {
        const int R = N - r0 < PC_RB ? N - r0 : PC_RB;
        const float* X = sl->obs + (size_t)r0 * O;
        affine(R, O, H1, X, O, n.W1, n.b1, h1, H1);
        for (int i = 0; i < R * H1; ++i) h1[i] = tanhf(h1[i]);
        affine(R, H1, H2, h1, H1, n.W2, n.b2, h2, H2);
        for (int i = 0; i < R * H2; ++i) h2[i] = tanhf(h2[i]);
        affine(R, H2, A, h2, H2, n.W3, n.b3, mu, A);
        float gs[PC_MAXA] = {0}, gks[PC_MAXA] = {0};
        for (int r = 0; r < R; ++r) {
            const int row = r0 + r;
            const float* ac = sl->act + (size_t)row * A;
            const float* om = sl->om + (size_t)row * A;
            float lp = 0.f, lpo = 0.f, kl = 0.f, z[PC_MAXA];
            for (int a = 0; a < A; ++a) {
                z[a] = (ac[a] - mu[r * A + a]) * e[a];
                const float zo = (ac[a] - om[a]) * eo[a];
                lp += -s[a] - 0.5f * z[a] * z[a];
                lpo += -sl->ols[a] - 0.5f * zo * zo;
                const float num = (om[a] - mu[r * A + a]) * (om[a] - mu[r * A + a]) + so2[a] - sn2[a], den = 2.f * sn2[a] + 1e-8f;
                kl += num / den + s[a] - sl->ols[a];
                dk[r * A + a] = -2.f * (om[a] - mu[r * A + a]) / den * invN;
                gks[a] += ((-2.f * sn2[a] * den - num * 4.f * sn2[a]) / (den * den) + 1.f) * invN;
            }
            kls += kl;
            const float adv = sl->adv[row];
            float c = 0.f;
            if (kind == PC_LOGLIK) {
                loss += -(lp - 0.5f * A * PC_LOG_2PI) * adv;
                c = -adv * invN;
            } else if (kind == PC_KL) {
                loss += kl;
            } else {
                const float rho = expf(lp - lpo);
                if (kind == PC_RATIO) {
                    loss += -rho * adv;
                    c = -adv * rho * invN;
                } else {
                    const float x = rho * adv, lo = 1.f - clip_eps, hi = 1.f + clip_eps;
                    const float y = (rho < lo ? lo : rho > hi ? hi : rho) * adv;
                    loss += -(x < y ? x : y);
                    c = (x <= y) ? -adv * rho * invN : 0.f;
                }
            }
            for (int a = 0; a < A; ++a) {
                dmu[r * A + a] = (kind == PC_KL) ? dk[r * A + a] : c * z[a] * e[a];
                gs[a] += (kind == PC_KL) ? 0.f : c * (z[a] * z[a] - 1.f);
            }
        }
        for (int pass = 0; pass < 2; ++pass) {
            float* g = pass == 0 ? grad : gkl;
            if (!g || (pass == 1 && kind == PC_KL)) continue;
            const float* dm = pass == 0 ? dmu : dk;
            float* gW1 = g; float* gb1 = gW1 + O * H1; float* gW2 = gb1 + H1; float* gb2 = gW2 + H1 * H2;
            float* gW3 = gb2 + H2; float* gb3 = gW3 + H2 * A; float* gS = gb3 + A;
            outer_acc(R, H2, A, h2, H2, dm, A, gW3, gb3);
            matmul_t(R, H2, A, dm, A, n.W3, dz2, H2, 0);
            for (int i = 0; i < R * H2; ++i) dz2[i] *= 1.f - h2[i] * h2[i];
            outer_acc(R, H1, H2, h1, H1, dz2, H2, gW2, gb2);
            matmul_t(R, H1, H2, dz2, H2, n.W2, dz1, H1, 0);
            for (int i = 0; i < R * H1; ++i) dz1[i] *= 1.f - h1[i] * h1[i];
            outer_acc(R, O, H1, X, O, dz1, H1, gW1, gb1);
            for (int a = 0; a < A; ++a) gS[a] += (pass == 0 ? (kind == PC_KL ? gks[a] : gs[a]) : gks[a]) * mask[a];
        }
    }
    *loss_out = (float)(loss * invN);
    *kl_out = (float)(kls * invN);
    if (kind == PC_KL && grad && gkl) memcpy(gkl, grad, sizeof(float) * NP);
}

/* out = (d^2 L / d theta^2) v of the inner objective (ratio | loglik) of ONE task: Pearlmutter R-operator */
static void hvp(const pc_dims* d, const float* th, const slab_t* sl, const float* v, int kind, int clip_ls, float* out) {
    const int O = d->O, A = d->A, H1 = d->H1, H2 = d->H2, N = sl->N, NP = n_params(d);
    const net_t n = net_of(d, th), q = net_of(d, v);
    const float min_ls = logf(1e-6f);
    float s[PC_MAXA], e[PC_MAXA], mask[PC_MAXA], Rs[PC_MAXA], eo[PC_MAXA];
    for (int a = 0; a < A; ++a) {
        const int clipped = clip_ls && n.s[a] < min_ls;
        s[a] = clipped ? min_ls : n.s[a];
        mask[a] = clipped ? 0.f : 1.f;
        e[a] = expf(-s[a]);
        Rs[a] = q.s[a] * mask[a];
        eo[a] = expf(-sl->ols[a]);
    }
    memset(out, 0, sizeof(float) * NP);
    float* oW1 = out; float* ob1 = oW1 + O * H1; float* oW2 = ob1 + H1; float* ob2 = oW2 + H1 * H2;
    float* oW3 = ob2 + H2; float* ob3 = oW3 + H2 * A; float* oS = ob3 + A;
    float h1[PC_RB * PC_MAXH], rh1[PC_RB * PC_MAXH], h2[PC_RB * PC_MAXH], rh2[PC_RB * PC_MAXH];
    float mu[PC_RB * PC_MAXA], rmu[PC_RB * PC_MAXA], dmu[PC_RB * PC_MAXA], rdmu[PC_RB * PC_MAXA];
    float dx[PC_RB * PC_MAXH], rdx[PC_RB * PC_MAXH], dz2[PC_RB * PC_MAXH], rdz2[PC_RB * PC_MAXH], rdz1[PC_RB * PC_MAXH];
    const float invN = 1.f / (float)N;
    for (int r0 = 0; r0 < N; r0 += PC_RB) {
        const int R = N - r0 < PC_RB ? N - r0 : PC_RB;
        const float* X = sl->obs + (size_t)r0 * O;
        affine(R, O, H1, X, O, n.W1, n.b1, h1, H1);
        affine(R, O, H1, X, O, q.W1, q.b1, rh1, H1);
        for (int i = 0; i < R * H1; ++i) { h1[i] = tanhf(h1[i]); rh1[i] *= 1.f - h1[i] * h1[i]; }
        affine(R, H1, H2, h1, H1, n.W2, n.b2, h2, H2);
        affine(R, H1, H2, h1, H1, q.W2, q.b2, rh2, H2);
        matmul_acc(R, H1, H2, rh1, H1, n.W2, rh2, H2);
        for (int i = 0; i < R * H2; ++i) { h2[i] = tanhf(h2[i]); rh2[i] *= 1.f - h2[i] * h2[i]; }
        affine(R, H2, A, h2, H2, n.W3, n.b3, mu, A);
        affine(R, H2, A, h2, H2, q.W3, q.b3, rmu, A);
        matmul_acc(R, H2, A, rh2, H2, n.W3, rmu, A);
        float rds[PC_MAXA] = {0};
        for (int r = 0; r < R; ++r) {
            const int row = r0 + r;
            const float* ac = sl->act + (size_t)row * A;
            const float* om = sl->om + (size_t)row * A;
            float lp = 0.f, lpo = 0.f, Rlp = 0.f, z[PC_MAXA];
            for (int a = 0; a < A; ++a) {
                z[a] = (ac[a] - mu[r * A + a]) * e[a];
                const float zo = (ac[a] - om[a]) * eo[a];
                lp += -s[a] - 0.5f * z[a] * z[a];
                lpo += -sl->ols[a] - 0.5f * zo * zo;
                Rlp += z[a] * e[a] * rmu[r * A + a] + (z[a] * z[a] - 1.f) * Rs[a];
            }
            const float adv = sl->adv[row];
            float c, Rc;
            if (kind == PC_RATIO) {
                c = -adv * expf(lp - lpo) * invN;
                Rc = c * Rlp;
            } else {
                c = -adv * invN;
                Rc = 0.f;
            }
            for (int a = 0; a < A; ++a) {
                const float Rz = -rmu[r * A + a] * e[a] - z[a] * Rs[a];
                dmu[r * A + a] = c * z[a] * e[a];
                rdmu[r * A + a] = Rc * z[a] * e[a] + c * (Rz * e[a] - z[a] * e[a] * Rs[a]);
                rds[a] += Rc * (z[a] * z[a] - 1.f) + c * 2.f * z[a] * Rz;
            }
        }
        /* R-backward */
        outer_acc(R, H2, A, rh2, H2, dmu, A, oW3, NULL);
        outer_acc(R, H2, A, h2, H2, rdmu, A, oW3, ob3);
        matmul_t(R, H2, A, dmu, A, n.W3, dx, H2, 0);
        matmul_t(R, H2, A, rdmu, A, n.W3, rdx, H2, 0);
        matmul_t(R, H2, A, dmu, A, q.W3, rdx, H2, 1);
        for (int i = 0; i < R * H2; ++i) {
            const float d1 = 1.f - h2[i] * h2[i];
            rdz2[i] = rdx[i] * d1 - 2.f * dx[i] * h2[i] * rh2[i];
            dz2[i] = dx[i] * d1;
        }
        outer_acc(R, H1, H2, rh1, H1, dz2, H2, oW2, NULL);
        outer_acc(R, H1, H2, h1, H1, rdz2, H2, oW2, ob2);
        matmul_t(R, H1, H2, dz2, H2, n.W2, dx, H1, 0);
        matmul_t(R, H1, H2, rdz2, H2, n.W2, rdx, H1, 0);
        matmul_t(R, H1, H2, dz2, H2, q.W2, rdx, H1, 1);
        for (int i = 0; i < R * H1; ++i) rdz1[i] = rdx[i] * (1.f - h1[i] * h1[i]) - 2.f * dx[i] * h1[i] * rh1[i];
        outer_acc(R, O, H1, X, O, rdz1, H1, oW1, ob1);
        for (int a = 0; a < A; ++a) oS[a] += rds[a] * mask[a];
    }
}

static slab_t task_slab(const pc_dims* d, int i, const float* obs, const float* act, const float* adv, const float* om,
                        const float* ols) {
    const size_t N = (size_t)d->P * d->T;
    slab_t s;
    s.N = (int)N;
    s.obs = obs + i * N * d->O; s.act = act + i * N * d->A; s.adv = adv + i * N; s.om = om + i * N * d->A; s.ols = ols + (size_t)i * d->A;
    return s;
}

/* MAMLAlgo._adapt: theta_out[i] = theta_in[i or shared] - alpha * grad L_i */
void pc_adapt(const pc_dims* d, const float* theta, int theta_per_task, const float* alpha, int inner_kind, const float* obs,
              const float* act, const float* adv, const float* om, const float* ols, float* theta_out) {
    const int NP = n_params(d);
#pragma omp parallel for schedule(dynamic)
    for (int i = 0; i < d->M; ++i) {
        float* g = (float*)malloc(sizeof(float) * NP);
        const float* th = theta + (theta_per_task ? (size_t)i * NP : 0);
        const slab_t s = task_slab(d, i, obs, act, adv, om, ols);
        float l, k;
        loss_grad(d, th, &s, inner_kind, 0, 0.f, &l, &k, g, NULL);
        for (int j = 0; j < NP; ++j) theta_out[(size_t)i * NP + j] = th[j] - alpha[j] * g[j];
        free(g);
    }
}

/* ProMP.build_graph at theta for K = 1 (pro_mp.py:67-163): stats = { loss, inner_kl, outer_kl }, grad = task-mean gradient */
void pc_meta_grad(const pc_dims* d, const float* theta, const float* alpha, float eta, float clip_eps, int inner_kind, int outer_kind,
                  int want_grad, const float* obs0, const float* act0, const float* adv0, const float* om0, const float* ols0,
                  const float* obs1, const float* act1, const float* adv1, const float* om1, const float* ols1, float* grad_out,
                  float* stats_out) {
    const int NP = n_params(d), M = d->M;
    double loss = 0.0, ikl = 0.0, okl = 0.0;
    float* gsum = (float*)calloc((size_t)M * NP, sizeof(float));
#pragma omp parallel for schedule(dynamic) reduction(+ : loss, ikl, okl)
    for (int i = 0; i < M; ++i) {
        float* buf = (float*)malloc(sizeof(float) * NP * 5);
        float *g0 = buf, *gk0 = buf + NP, *th1 = buf + 2 * NP, *lam = buf + 3 * NP, *hv = buf + 4 * NP;
        const slab_t s0 = task_slab(d, i, obs0, act0, adv0, om0, ols0), s1 = task_slab(d, i, obs1, act1, adv1, om1, ols1);
        float l0, k0, l1, k1;
        loss_grad(d, theta, &s0, inner_kind, 1, 0.f, &l0, &k0, g0, gk0);
        for (int j = 0; j < NP; ++j) th1[j] = theta[j] - alpha[j] * g0[j];
        loss_grad(d, th1, &s1, outer_kind, 0, clip_eps, &l1, &k1, want_grad ? lam : NULL, NULL);
        loss += l1; ikl += k0; okl += k1;
        if (want_grad) {
            for (int j = 0; j < NP; ++j) g0[j] = alpha[j] * lam[j];          /* v = alpha * lam */
            hvp(d, theta, &s0, g0, inner_kind, 1, hv);
            for (int j = 0; j < NP; ++j) gsum[(size_t)i * NP + j] = lam[j] - hv[j] + eta * gk0[j];
        }
        free(buf);
    }
    if (want_grad) {
        for (int j = 0; j < NP; ++j) {
            float t = 0.f;
            for (int i = 0; i < M; ++i) t += gsum[(size_t)i * NP + j];
            grad_out[j] = t / (float)M;
        }
    }
    free(gsum);
    stats_out[1] = (float)(ikl / M);
    stats_out[2] = (float)(okl / M);
    stats_out[0] = (float)(loss / M) + eta * stats_out[1];
}

void pc_adam(int n, float* theta, float* m, float* v, long long* t, const float* grad, float lr) {
    *t += 1;
    const double lr_t = (double)lr * sqrt(1.0 - pow(0.999, (double)*t)) / (1.0 - pow(0.9, (double)*t));
    for (int j = 0; j < n; ++j) {
        m[j] = 0.9f * m[j] + 0.1f * grad[j];
        v[j] = 0.999f * v[j] + 0.001f * grad[j] * grad[j];
        theta[j] -= (float)lr_t * m[j] / (sqrtf(v[j]) + 1e-8f);
    }
}

/* MetaSampleProcessor.process_samples with LinearFeatureBaseline, float64 like the reference; adv_out float32 [rows] */
void pc_process_samples(const pc_dims* d, const float* obs, const float* rew, double gamma, double lam, double reg, int normalize,
                        float* adv_out, double* ret_out) {
    const int P = d->P, T = d->T, O = d->O, D = 2 * O + 4;
    const size_t N = (size_t)P * T;
#pragma omp parallel for schedule(dynamic)
    for (int i = 0; i < d->M; ++i) {
        const float* ob = obs + i * N * O;
        const float* rw = rew + i * N;
        double* ret = ret_out + i * N;
        double* G = (double*)calloc((size_t)D * D + 3 * D, sizeof(double));
        double *rhs = G + (size_t)D * D, *w = rhs + D, *phi = w + D;
        double* adv = (double*)malloc(sizeof(double) * N * 2);
        double* bl = adv + N;
        for (int p = 0; p < P; ++p) {               /* returns: utils.discount_cumsum */
            double run = 0.0;
            for (int t = T - 1; t >= 0; --t) { run = rw[p * T + t] + gamma * run; ret[p * T + t] = run; }
        }
        for (size_t r = 0; r < N; ++r) {            /* normal equations Phi^T Phi, Phi^T R */
            const double tt = (double)(r % T) / 100.0;
            for (int k = 0; k < O; ++k) {
                double c = ob[r * O + k];
                c = c < -10.0 ? -10.0 : c > 10.0 ? 10.0 : c;
                phi[k] = c; phi[O + k] = c * c;
            }
            phi[2 * O] = tt; phi[2 * O + 1] = tt * tt; phi[2 * O + 2] = tt * tt * tt; phi[2 * O + 3] = 1.0;
            for (int a = 0; a < D; ++a) {
                const double pa = phi[a];
                for (int b = 0; b < D; ++b) G[a * D + b] += pa * phi[b];
                rhs[a] += pa * ret[r];
            }
        }
        double* L = (double*)malloc(sizeof(double) * D * D);
        double rg = reg;
        for (int attempt = 0; attempt < 5; ++attempt) {     /* Cholesky solve of (G + reg I) w = rhs; NaN -> reg *= 10 */
            memcpy(L, G, sizeof(double) * D * D);
            for (int a = 0; a < D; ++a) L[a * D + a] += rg;
            for (int j = 0; j < D; ++j) {
                double s = L[j * D + j];
                for (int k = 0; k < j; ++k) s -= L[j * D + k] * L[j * D + k];
                const double piv = sqrt(s);
                L[j * D + j] = piv;
                for (int a = j + 1; a < D; ++a) {
                    double t2 = L[a * D + j];
                    for (int k = 0; k < j; ++k) t2 -= L[a * D + k] * L[j * D + k];
                    L[a * D + j] = t2 / piv;
                }
            }
            for (int a = 0; a < D; ++a) {
                double t2 = rhs[a];
                for (int k = 0; k < a; ++k) t2 -= L[a * D + k] * w[k];
                w[a] = t2 / L[a * D + a];
            }
            for (int a = D - 1; a >= 0; --a) {
                double t2 = w[a];
                for (int k = a + 1; k < D; ++k) t2 -= L[k * D + a] * w[k];
                w[a] = t2 / L[a * D + a];
            }
            int bad = 0;
            for (int a = 0; a < D; ++a) bad |= (w[a] != w[a]);
            if (!bad) break;
            rg *= 10.0;
        }
        free(L);
        for (size_t r = 0; r < N; ++r) {            /* baseline = Phi w */
            const double tt = (double)(r % T) / 100.0;
            double b = w[2 * O] * tt + w[2 * O + 1] * tt * tt + w[2 * O + 2] * tt * tt * tt + w[2 * O + 3];
            for (int k = 0; k < O; ++k) {
                double c = ob[r * O + k];
                c = c < -10.0 ? -10.0 : c > 10.0 ? 10.0 : c;
                b += w[k] * c + w[O + k] * c * c;
            }
            bl[r] = b;
        }
        double sum = 0.0, sq = 0.0;
        for (int p = 0; p < P; ++p) {               /* GAE: samplers/base.py:151-162 */
            double run = 0.0;
            for (int t = T - 1; t >= 0; --t) {
                const size_t r = (size_t)p * T + t;
                const double delta = rw[r] + gamma * (t + 1 < T ? bl[r + 1] : 0.0) - bl[r];
                run = delta + gamma * lam * run;
                adv[r] = run;
            }
        }
        for (size_t r = 0; r < N; ++r) sum += adv[r];
        const double mean = sum / (double)N;
        for (size_t r = 0; r < N; ++r) sq += (adv[r] - mean) * (adv[r] - mean);
        const double sd = sqrt(sq / (double)N);
        for (size_t r = 0; r < N; ++r) adv_out[i * N + r] = (float)(normalize ? (adv[r] - mean) / (sd + 1e-8) : adv[r]);
        free(adv);
        free(G);
    }
}
