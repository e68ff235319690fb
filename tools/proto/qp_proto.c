/* Prototype playground for QP algorithm changes (not shipped): includes the oracle source. */
#include "../../oracle/nmpc_oracle.c"

typedef struct proto_opts {
    int method;        /* 0 IPM, 1 PDAS(+IPM fallback), 2 IPM+polish */
    double rho;
    int refine;        /* multiplier refinements at the end */
    int pdas_max;
    double tol_p, tol_d;
    int warm_ws;       /* use ws_in as the initial working set */
    double mu_pol; int need_stable;
    double warm_eps; int warm_tries; int warm_gate;
} proto_opts;

typedef struct proto_stats { int status, n_factor, n_refine, n_ipm, fallback, n_active; } proto_stats;

/* ws: N*4 ints (-1 lower, 0 free, +1 upper), in/out */
static int qp_pdas(const orc_config *cfg, const orc_qp *qp, const proto_opts *o, double *dx, double *du,
                   int *ws, proto_stats *st)
{
    const int N = qp->N;
    const size_t nb = (size_t)N * NB;
    double *mem = (double *)calloc((size_t)(N + 1) * NZ * 2 + nb * 4 + (size_t)N * (64 + 8 + 6 + 8 + 12 + 4 + 2), sizeof(double));
    double *z = mem, *dz = z + (N + 1) * NZ;
    double *lo = dz + (N + 1) * NZ, *up = lo + nb, *msk = up + nb, *s = msk + nb;
    double *Hd = s + nb, *G = Hd + (size_t)N * 64, *rb = G + (size_t)N * 8, *dG = rb + (size_t)N * 6;
    ric_ws w; w.N = N; w.K = dG + (size_t)N * 8; w.Hinv = w.K + (size_t)N * 12; w.kff = w.Hinv + (size_t)N * 4;
    for (int i = 0; i < NX; ++i) z[i] = qp->dx0[i];
    for (int k = 0; k < N; ++k)
        for (int j = 0; j < NB; ++j) {
            const size_t q = (size_t)k * NB + j;
            if (j < 2) { lo[q] = qp->lu[k * 2 + j]; up[q] = qp->uu[k * 2 + j]; msk[q] = 1.0; }
            else { lo[q] = qp->lx[k * 2 + j - 2]; up[q] = qp->ux[k * 2 + j - 2]; msk[q] = (k >= 1) ? 1.0 : 0.0; }
            if (!o->warm_ws || msk[q] == 0.0) ws[q] = 0;
        }
    int status = 4, it;
    for (it = 0; it < o->pdas_max; ++it) {
        for (int k = 0; k < N; ++k) {
            const double *A = qp->A + k * 36, *B = qp->B + k * 12, *H = qp->H + k * 64;
            const double *zk = z + k * NZ, *zn = z + (k + 1) * NZ;
            for (int i = 0; i < NX; ++i) {
                double acc = qp->b[k * NX + i] - zn[i];
                for (int c = 0; c < NX; ++c) acc += A[i * NX + c] * zk[c];
                for (int a = 0; a < NU; ++a) acc += B[i * NU + a] * zk[NX + a];
                rb[k * NX + i] = acc;
            }
            memcpy(Hd + k * 64, H, sizeof(double) * 64);
            for (int i = 0; i < NZ; ++i) {
                double acc = qp->g[k * NZ + i];
                for (int j = 0; j < NZ; ++j) acc += H[i * NZ + j] * zk[j];
                G[k * NZ + i] = acc;
            }
            for (int j = 0; j < NB; ++j) {
                const size_t q = (size_t)k * NB + j;
                const int zi = BIDX[j];
                if (ws[q] != 0) {
                    const double bd = ws[q] < 0 ? lo[q] : up[q];
                    Hd[k * 64 + zi * NZ + zi] += o->rho;
                    G[k * NZ + zi] += s[q] + o->rho * (zk[zi] - bd);
                }
            }
        }
        riccati_solve(qp, &w, Hd, G, rb, 1, dz);
        st->n_factor++;
        for (int k = 0; k <= N; ++k) for (int i = 0; i < NZ; ++i) z[k * NZ + i] += dz[k * NZ + i];
        for (size_t q = 0; q < nb; ++q) if (ws[q] != 0) {
            const int k = (int)(q / NB), zi = BIDX[q % NB];
            s[q] += o->rho * (z[k * NZ + zi] - (ws[q] < 0 ? lo[q] : up[q]));
        }
        /* new working set */
        int changed = 0, nact = 0;
        for (size_t q = 0; q < nb; ++q) {
            if (msk[q] == 0.0) continue;
            const int k = (int)(q / NB), zi = BIDX[q % NB];
            const double v = z[k * NZ + zi];
            if (ws[q] == 0) {
                if (v < lo[q] - o->tol_p) { ws[q] = -1; s[q] = 0.0; changed = 1; }
                else if (v > up[q] + o->tol_p) { ws[q] = 1; s[q] = 0.0; changed = 1; }
            } else if (ws[q] < 0) {
                if (s[q] > o->tol_d) { ws[q] = 0; s[q] = 0.0; changed = 1; }
            } else {
                if (s[q] < -o->tol_d) { ws[q] = 0; s[q] = 0.0; changed = 1; }
            }
            if (ws[q] != 0) nact++;
        }
        st->n_active = nact;
        if (!changed) {
            if (nact > 0)
                for (int r = 0; r < o->refine; ++r) {
                    memset(dG, 0, sizeof(double) * (size_t)N * 8);
                    for (size_t q = 0; q < nb; ++q) if (ws[q] != 0) {
                        const int k = (int)(q / NB), zi = BIDX[q % NB];
                        dG[k * NZ + zi] = o->rho * (z[k * NZ + zi] - (ws[q] < 0 ? lo[q] : up[q]));
                    }
                    riccati_solve(qp, &w, Hd, dG, rb, 0, dz);
                    st->n_refine++;
                    for (int k = 0; k <= N; ++k) for (int i = 0; i < NZ; ++i) z[k * NZ + i] += dz[k * NZ + i];
                    for (size_t q = 0; q < nb; ++q) if (ws[q] != 0) {
                        const int k = (int)(q / NB), zi = BIDX[q % NB];
                        s[q] += o->rho * (z[k * NZ + zi] - (ws[q] < 0 ? lo[q] : up[q]));
                    }
                }
            status = 0; ++it;
            break;
        }
    }
    for (int k = 0; k <= N; ++k) for (int i = 0; i < NX; ++i) dx[k * NX + i] = z[k * NZ + i];
    for (int k = 0; k < N; ++k) { du[k * NU] = z[k * NZ + 6]; du[k * NU + 1] = z[k * NZ + 7]; }
    free(mem);
    return status;
}


/* one active-set solve from z (in/out) for the working set ws: returns 1 if the result satisfies the KKT
   conditions of the QP (multiplier signs, primal feasibility), 0 otherwise; s = signed multipliers */
static __thread int g_hint;   /* in: mismatches of the previous call (-1 unknown); out: of this one */
typedef struct pol_ws { double *Hd, *G, *rb, *dG, *dz, *s; ric_ws w; } pol_ws;
static int polish(const orc_qp *qp, const proto_opts *o, const double *lo, const double *up, const double *msk,
                  const int *ws, double *z, pol_ws *P, proto_stats *st, double *viol_out)
{
    const int N = qp->N;
    const size_t nb = (size_t)N * NB;
    double *Hd = P->Hd, *G = P->G, *rb = P->rb, *dG = P->dG, *dz = P->dz, *s = P->s;
    for (int k = 0; k < N; ++k) {
        const double *A = qp->A + k * 36, *B = qp->B + k * 12, *H = qp->H + k * 64;
        const double *zk = z + k * NZ, *zn = z + (k + 1) * NZ;
        for (int i = 0; i < NX; ++i) {
            double acc = qp->b[k * NX + i] - zn[i];
            for (int c = 0; c < NX; ++c) acc += A[i * NX + c] * zk[c];
            for (int a = 0; a < NU; ++a) acc += B[i * NU + a] * zk[NX + a];
            rb[k * NX + i] = acc;
        }
        memcpy(Hd + k * 64, H, sizeof(double) * 64);
        for (int i = 0; i < NZ; ++i) {
            double acc = qp->g[k * NZ + i];
            for (int j = 0; j < NZ; ++j) acc += H[i * NZ + j] * zk[j];
            G[k * NZ + i] = acc;
        }
        for (int j = 0; j < NB; ++j) {
            const size_t q = (size_t)k * NB + j;
            const int zi = BIDX[j];
            if (ws[q] != 0) {
                const double bd = ws[q] < 0 ? lo[q] : up[q];
                Hd[k * 64 + zi * NZ + zi] += o->rho;
                G[k * NZ + zi] += s[q] + o->rho * (zk[zi] - bd);
            }
        }
    }
    riccati_solve(qp, &P->w, Hd, G, rb, 1, dz);
    st->n_factor++;
    for (int k = 0; k <= N; ++k) for (int i = 0; i < NZ; ++i) z[k * NZ + i] += dz[k * NZ + i];
    for (int r = 0; r <= o->refine; ++r) {
        for (size_t q = 0; q < nb; ++q) if (ws[q] != 0) {
            const int k = (int)(q / NB), zi = BIDX[q % NB];
            s[q] += o->rho * (z[k * NZ + zi] - (ws[q] < 0 ? lo[q] : up[q]));
        }
        if (r == o->refine) break;
        memset(dG, 0, sizeof(double) * (size_t)N * 8);
        for (size_t q = 0; q < nb; ++q) if (ws[q] != 0) {
            const int k = (int)(q / NB), zi = BIDX[q % NB];
            dG[k * NZ + zi] = o->rho * (z[k * NZ + zi] - (ws[q] < 0 ? lo[q] : up[q]));
        }
        riccati_solve(qp, &P->w, Hd, dG, rb, 0, dz);
        st->n_refine++;
        for (int k = 0; k <= N; ++k) for (int i = 0; i < NZ; ++i) z[k * NZ + i] += dz[k * NZ + i];
    }
    int ok = 1;
    double viol = 0.0;
    for (size_t q = 0; q < nb; ++q) {
        if (msk[q] == 0.0) continue;
        const int k = (int)(q / NB), zi = BIDX[q % NB];
        const double v = z[k * NZ + zi];
        if (ws[q] == 0) { if (v < lo[q] - o->tol_p || v > up[q] + o->tol_p) ok = 0; viol = fmax(viol, fmax(lo[q] - v, v - up[q])); }
        else if (ws[q] < 0) { if (s[q] > o->tol_d) ok = 0; viol = fmax(viol, s[q]); }
        else { if (s[q] < -o->tol_d) ok = 0; viol = fmax(viol, -s[q]); }
    }
    if (viol_out) *viol_out = viol;
    return ok;
}

/* IPM as orc_qp_ipm, with an active-set polish attempted once mu <= mu_pol */
static int qp_ipm_polish(const orc_config *cfg, const orc_qp *qp, const proto_opts *o, double mu_pol, int need_stable,
                         double *dx, double *du, proto_stats *st)
{
    const int N = qp->N;
    const size_t nb = (size_t)N * NB;
    double *mem = (double *)calloc((size_t)(N + 1) * NZ * 5 + nb * 14 + (size_t)N * (64 + 8 + 6 + 8 + 12 + 4 + 2) * 2, sizeof(double));
    double *z = mem, *dz = z + (N + 1) * NZ, *dzc = dz + (N + 1) * NZ, *zp = dzc + (N + 1) * NZ, *dzp = zp + (N + 1) * NZ;
    double *tl = dzp + (N + 1) * NZ, *tu = tl + nb, *ll = tu + nb, *lu = ll + nb;
    double *dtl = lu + nb, *dtu = dtl + nb, *dll = dtu + nb, *dlu = dll + nb;
    double *lo = dlu + nb, *up = lo + nb, *msk = up + nb, *rtl = msk + nb, *rtu = rtl + nb, *sp = rtu + nb;
    double *Hd = sp + nb, *G = Hd + (size_t)N * 64, *rb = G + (size_t)N * 8, *dG = rb + (size_t)N * 6;
    ric_ws w; w.N = N; w.K = dG + (size_t)N * 8; w.Hinv = w.K + (size_t)N * 12; w.kff = w.Hinv + (size_t)N * 4;
    pol_ws P;
    P.Hd = w.kff + (size_t)N * 2; P.G = P.Hd + (size_t)N * 64; P.rb = P.G + (size_t)N * 8; P.dG = P.rb + (size_t)N * 6;
    P.w.N = N; P.w.K = P.dG + (size_t)N * 8; P.w.Hinv = P.w.K + (size_t)N * 12; P.w.kff = P.w.Hinv + (size_t)N * 4;
    P.dz = dzp; P.s = sp;
    int *ws = (int *)calloc(nb * 2, sizeof(int)), *wsprev = ws + nb;
    for (size_t q = 0; q < nb; ++q) wsprev[q] = 99;

    int m = 0, status = 4, it = 0;
    for (int i = 0; i < NX; ++i) z[i] = qp->dx0[i];
    for (int k = 0; k < N; ++k)
        for (int j = 0; j < NB; ++j) {
            const size_t q = (size_t)k * NB + j;
            if (j < 2) { lo[q] = qp->lu[k * 2 + j]; up[q] = qp->uu[k * 2 + j]; msk[q] = 1.0; }
            else { lo[q] = qp->lx[k * 2 + j - 2]; up[q] = qp->ux[k * 2 + j - 2]; msk[q] = (k >= 1) ? 1.0 : 0.0; }
            if (msk[q] != 0.0) {
                const double wv = z[k * NZ + BIDX[j]];
                tl[q] = fmax(wv - lo[q], cfg->ipm_thr0);
                tu[q] = fmax(up[q] - wv, cfg->ipm_thr0);
                ll[q] = cfg->ipm_mu0 / tl[q];
                lu[q] = cfg->ipm_mu0 / tu[q];
                m += 2;
            } else { tl[q] = tu[q] = 1.0; ll[q] = lu[q] = 0.0; lo[q] = -1.0; up[q] = 1.0; }
        }
    double lin_scale = 1.0, mu = 0.0;
    int *ws0 = (int *)calloc(nb, sizeof(int));
    for (size_t q = 0; q < nb; ++q) {
        ws0[q] = 0;
        if (msk[q] == 0.0) continue;
        if (lo[q] >= -o->warm_eps) ws0[q] = -1; else if (up[q] <= o->warm_eps) ws0[q] = 1;
    }
    const int hint_in = g_hint;
    g_hint = 1000;
    if (o->warm_ws && hint_in >= 0 && hint_in <= o->warm_gate) {
        int nact = 0;
        for (size_t q = 0; q < nb; ++q) {
            ws[q] = 0; sp[q] = 0.0;
            if (msk[q] == 0.0) continue;
            if (lo[q] >= -o->warm_eps) ws[q] = -1; else if (up[q] <= o->warm_eps) ws[q] = 1;
            if (ws[q]) nact++;
        }
        memcpy(zp, z, sizeof(double) * (size_t)(N + 1) * NZ);
        double viol;
        st->n_active = nact;
        int okp = polish(qp, o, lo, up, msk, ws, zp, &P, st, &viol);
        for (int tr = 0; !okp && tr < o->warm_tries; ++tr) {
            for (size_t q = 0; q < nb; ++q) {
                if (msk[q] == 0.0) continue;
                const int k = (int)(q / NB), zi = BIDX[q % NB];
                const double v = zp[k * NZ + zi];
                if (ws[q] == 0) {
                    if (v < lo[q] - o->tol_p) { ws[q] = -1; sp[q] = 0.0; }
                    else if (v > up[q] + o->tol_p) { ws[q] = 1; sp[q] = 0.0; }
                } else if (ws[q] < 0) { if (sp[q] > o->tol_d) { ws[q] = 0; sp[q] = 0.0; } }
                else { if (sp[q] < -o->tol_d) { ws[q] = 0; sp[q] = 0.0; } }
            }
            okp = polish(qp, o, lo, up, msk, ws, zp, &P, st, &viol);
        }
        if (okp) {
            int mm = 0;
            for (size_t q = 0; q < nb; ++q) mm += ws[q] != ws0[q];
            g_hint = mm;
            free(ws0);
            memcpy(z, zp, sizeof(double) * (size_t)(N + 1) * NZ);
            for (int k = 0; k <= N; ++k) for (int i = 0; i < NX; ++i) dx[k * NX + i] = z[k * NZ + i];
            for (int k = 0; k < N; ++k) { du[k * NU] = z[k * NZ + 6]; du[k * NU + 1] = z[k * NZ + 7]; }
            st->n_ipm = 0;
            free(mem); free(ws);
            return 0;
        }
        st->fallback += 100;
    }
    for (it = 0;; ++it) {
        mu = 0.0;
        for (size_t q = 0; q < nb; ++q) mu += msk[q] * (tl[q] * ll[q] + tu[q] * lu[q]);
        mu /= (double)m;
        if (!(mu == mu)) { status = 1; break; }
        double cmax = 0.0;
        for (size_t q = 0; q < nb; ++q)
            cmax = fmax(cmax, fmax(fmin(tl[q], ll[q]), fmin(tu[q], lu[q])));
        if (it > 0 && mu <= cfg->ipm_tol_mu && lin_scale <= cfg->ipm_tol_lin && cmax <= cfg->ipm_tol_comp) { status = 0; break; }
        if (it >= cfg->ipm_max_iter) { status = (mu <= 1e-8 && lin_scale <= 1e-8) ? 0 : 4; break; }
        if (cfg->ipm_stall_iter > 0 && it >= cfg->ipm_stall_iter && lin_scale > cfg->ipm_stall_lin) { status = 4; break; }

        if (it > 0 && mu <= mu_pol) {
            int same = 1, nact = 0;
            for (size_t q = 0; q < nb; ++q) {
                int c = 0;
                if (msk[q] != 0.0) { if (tl[q] < ll[q]) c = -1; else if (tu[q] < lu[q]) c = 1; }
                ws[q] = c;
                if (c != wsprev[q]) same = 0;
                if (c) nact++;
            }
            int attempt = need_stable ? same : !same;   /* need_stable: polish when the classification repeats; else when it is new */
            if (need_stable == 2) attempt = 1;
            memcpy(wsprev, ws, nb * sizeof(int));
            if (attempt) {
                memcpy(zp, z, sizeof(double) * (size_t)(N + 1) * NZ);
                for (size_t q = 0; q < nb; ++q) sp[q] = ws[q] < 0 ? -ll[q] : ws[q] > 0 ? lu[q] : 0.0;
                double viol;
                st->n_active = nact;
                int okp = polish(qp, o, lo, up, msk, ws, zp, &P, st, &viol);
                for (int tr = 0; !okp && tr < o->pdas_max; ++tr) {
                    /* primal-dual active-set update from the failed polish */
                    for (size_t q = 0; q < nb; ++q) {
                        if (msk[q] == 0.0) continue;
                        const int k = (int)(q / NB), zi = BIDX[q % NB];
                        const double v = zp[k * NZ + zi];
                        if (ws[q] == 0) {
                            if (v < lo[q] - o->tol_p) { ws[q] = -1; sp[q] = 0.0; }
                            else if (v > up[q] + o->tol_p) { ws[q] = 1; sp[q] = 0.0; }
                        } else if (ws[q] < 0) { if (sp[q] > o->tol_d) { ws[q] = 0; sp[q] = 0.0; } }
                        else { if (sp[q] < -o->tol_d) { ws[q] = 0; sp[q] = 0.0; } }
                    }
                    okp = polish(qp, o, lo, up, msk, ws, zp, &P, st, &viol);
                }
                if (okp) {
                    int mm = 0;
                    for (size_t q = 0; q < nb; ++q) mm += ws[q] != ws0[q];
                    g_hint = mm;
                    memcpy(z, zp, sizeof(double) * (size_t)(N + 1) * NZ);
                    status = 0;
                    break;
                }
                st->fallback++;
                if (need_stable == 1) for (size_t q = 0; q < nb; ++q) wsprev[q] = 99;   /* do not retry the same set at once */
            }
        }

        for (int k = 0; k < N; ++k) {
            const double *A = qp->A + k * 36, *B = qp->B + k * 12, *H = qp->H + k * 64;
            const double *zk = z + k * NZ, *zn = z + (k + 1) * NZ;
            for (int i = 0; i < NX; ++i) {
                double acc = qp->b[k * NX + i] - zn[i];
                for (int c = 0; c < NX; ++c) acc += A[i * NX + c] * zk[c];
                for (int a = 0; a < NU; ++a) acc += B[i * NU + a] * zk[NX + a];
                rb[k * NX + i] = acc;
            }
            memcpy(Hd + k * 64, H, sizeof(double) * 64);
            for (int i = 0; i < NZ; ++i) {
                double acc = qp->g[k * NZ + i];
                for (int j = 0; j < NZ; ++j) acc += H[i * NZ + j] * zk[j];
                G[k * NZ + i] = acc;
            }
            for (int j = 0; j < NB; ++j) {
                const size_t q = (size_t)k * NB + j;
                const int zi = BIDX[j];
                const double wv = zk[zi];
                rtl[q] = msk[q] * (wv - lo[q] - tl[q]);
                rtu[q] = msk[q] * (up[q] - wv - tu[q]);
                Hd[k * 64 + zi * NZ + zi] += ll[q] / tl[q] + lu[q] / tu[q];
                G[k * NZ + zi] += ll[q] / tl[q] * rtl[q] - lu[q] / tu[q] * rtu[q];
            }
        }
        riccati_solve(qp, &w, Hd, G, rb, 1, dz);
        double a_aff = 1.0;
        for (int k = 0; k < N; ++k)
            for (int j = 0; j < NB; ++j) {
                const size_t q = (size_t)k * NB + j;
                const double dw = dz[k * NZ + BIDX[j]];
                dtl[q] = msk[q] * (dw + rtl[q]);
                dtu[q] = msk[q] * (-dw + rtu[q]);
                dll[q] = -ll[q] - ll[q] / tl[q] * dtl[q];
                dlu[q] = -lu[q] - lu[q] / tu[q] * dtu[q];
                if (dtl[q] < 0.0) a_aff = fmin(a_aff, -tl[q] / dtl[q]);
                if (dtu[q] < 0.0) a_aff = fmin(a_aff, -tu[q] / dtu[q]);
                if (dll[q] < 0.0) a_aff = fmin(a_aff, -ll[q] / dll[q]);
                if (dlu[q] < 0.0) a_aff = fmin(a_aff, -lu[q] / dlu[q]);
            }
        double mu_aff = 0.0;
        for (size_t q = 0; q < nb; ++q)
            mu_aff += msk[q] * ((tl[q] + a_aff * dtl[q]) * (ll[q] + a_aff * dll[q]) +
                                (tu[q] + a_aff * dtu[q]) * (lu[q] + a_aff * dlu[q]));
        mu_aff /= (double)m;
        const double sr = mu_aff / mu;
        const double smu = sr * sr * sr * mu;
        const double csc = fmin(1.0, 4.0 * a_aff);
        memset(dG, 0, sizeof(double) * (size_t)N * 8);
        for (int k = 0; k < N; ++k)
            for (int j = 0; j < NB; ++j) {
                const size_t q = (size_t)k * NB + j;
                const double cl = smu - csc * dtl[q] * dll[q], cu = smu - csc * dtu[q] * dlu[q];
                dG[k * NZ + BIDX[j]] = msk[q] * (-cl / tl[q] + cu / tu[q]);
            }
        riccati_solve(qp, &w, Hd, dG, rb, 0, dzc);
        double a_max = 1e300;
        for (int k = 0; k <= N; ++k)
            for (int i = 0; i < NZ; ++i) dz[k * NZ + i] += dzc[k * NZ + i];
        for (int k = 0; k < N; ++k)
            for (int j = 0; j < NB; ++j) {
                const size_t q = (size_t)k * NB + j;
                const double dw = dz[k * NZ + BIDX[j]];
                const double cl = smu - csc * dtl[q] * dll[q], cu = smu - csc * dtu[q] * dlu[q];
                const double ntl = msk[q] * (dw + rtl[q]), ntu = msk[q] * (-dw + rtu[q]);
                dll[q] = msk[q] * (cl / tl[q] - ll[q] - ll[q] / tl[q] * ntl);
                dlu[q] = msk[q] * (cu / tu[q] - lu[q] - lu[q] / tu[q] * ntu);
                dtl[q] = ntl;
                dtu[q] = ntu;
                if (dtl[q] < 0.0) a_max = fmin(a_max, -tl[q] / dtl[q]);
                if (dtu[q] < 0.0) a_max = fmin(a_max, -tu[q] / dtu[q]);
                if (dll[q] < 0.0) a_max = fmin(a_max, -ll[q] / dll[q]);
                if (dlu[q] < 0.0) a_max = fmin(a_max, -lu[q] / dlu[q]);
            }
        const double tau = fmax(cfg->ipm_tau, 1.0 - mu_aff);
        const double alpha = fmin(1.0, tau * a_max);
        for (int k = 0; k <= N; ++k)
            for (int i = 0; i < NZ; ++i) z[k * NZ + i] += alpha * dz[k * NZ + i];
        for (size_t q = 0; q < nb; ++q) {
            tl[q] += alpha * dtl[q]; tu[q] += alpha * dtu[q];
            ll[q] += alpha * dll[q]; lu[q] += alpha * dlu[q];
        }
        lin_scale *= (1.0 - alpha);
    }
    for (int k = 0; k <= N; ++k) for (int i = 0; i < NX; ++i) dx[k * NX + i] = z[k * NZ + i];
    for (int k = 0; k < N; ++k) { du[k * NU] = z[k * NZ + 6]; du[k * NU + 1] = z[k * NZ + 7]; }
    st->n_ipm = it;
    free(mem); free(ws); free(ws0);
    return status;
}

void proto_rti_batch(const orc_config *c, const proto_opts *o, int Bn, const double *x0, const double *p,
                     double *xtraj, double *utraj, int *wsall, proto_stats *stats, int cold)
{
    const int N = c->N;
    const size_t sx = (size_t)(N + 1) * NX, su = (size_t)N * NU;
#pragma omp parallel for schedule(dynamic, 8)
    for (int i = 0; i < Bn; ++i) {
        double *xt = xtraj + i * sx, *ut = utraj + i * su;
        proto_stats st; memset(&st, 0, sizeof(st));
        if (cold) orc_init_guess(c, x0 + (size_t)i * NX, xt, ut);
        lin_ws w;
        lin_alloc(&w, N);
        linearise(c, p + (size_t)i * ORC_NP, xt, ut, &w);
        double dx0[NX];
        for (int m = 0; m < NX; ++m) dx0[m] = x0[(size_t)i * NX + m] - xt[m];
        orc_qp qp = {N, w.A, w.B, w.b, w.H, w.g, w.lu, w.uu, w.lx, w.ux, dx0};
        int qs = 4;
        if (o->method == 1) {
            qs = qp_pdas(c, &qp, o, w.dx, w.du, wsall + (size_t)i * N * NB, &st);
            if (qs != 0) st.fallback = 1;
        }
        if (o->method == 2) {
            qs = qp_ipm_polish(c, &qp, o, o->mu_pol, o->need_stable, w.dx, w.du, &st);
        } else if (qs != 0) {
            int iters = 0;
            qs = orc_qp_ipm(c, &qp, w.dx, w.du, w.lam, w.pi, &iters, NULL);
            st.n_ipm = iters;
        }
        st.status = qs;
        if (qs != 1) {
            for (size_t m = 0; m < sx; ++m) xt[m] += w.dx[m];
            for (size_t m = 0; m < su; ++m) ut[m] += w.du[m];
        }
        stats[i] = st;
    }
}

void proto_closed_loop(const orc_config *c, const proto_opts *o, const double *track, int nT, int Bn,
                       double *wsall, const double *pv, int steps, double dtp, proto_stats *stats /* [steps][Bn] */)
{
    const int N = c->N;
    orc_config pc = *c; pc.dt = dtp;
#pragma omp parallel for schedule(dynamic, 4)
    for (int i = 0; i < Bn; ++i) {
        double *wst = wsall + (size_t)i * 6;
        double *xt = (double *)calloc((size_t)(N + 1) * NX + (size_t)N * NU, sizeof(double)), *ut = xt + (size_t)(N + 1) * NX;
        int *wsd = (int *)calloc((size_t)N * NB, sizeof(int));
        int have = 0, hint = -1;
        for (int s = 0; s < steps; ++s) {
            double coef[4];
            const double rot = orc_calc_coeffs(track, nT, wst, coef);
            double xs[6] = {0, 0, rot, wst[3], wst[4], wst[5]};
            double p[6] = {coef[0], coef[1], coef[2], coef[3], pv[i * 2], pv[i * 2 + 1]};
            proto_opts ol = *o;
            if (!have) { orc_init_guess(c, xs, xt, ut); ol.warm_ws = 0; }
            proto_stats st; memset(&st, 0, sizeof(st));
            lin_ws w;
            lin_alloc(&w, N);
            linearise(c, p, xt, ut, &w);
            double dx0[NX];
            for (int m = 0; m < NX; ++m) dx0[m] = xs[m] - xt[m];
            orc_qp qp = {N, w.A, w.B, w.b, w.H, w.g, w.lu, w.uu, w.lx, w.ux, dx0};
            int qs;
            if (ol.method == 2) { g_hint = have ? hint : -1; qs = qp_ipm_polish(c, &qp, &ol, ol.mu_pol, ol.need_stable, w.dx, w.du, &st); hint = g_hint; }
            else { int iters = 0; qs = orc_qp_ipm(c, &qp, w.dx, w.du, w.lam, w.pi, &iters, NULL); st.n_ipm = iters; }
            st.status = qs;
            if (qs != 1) {
                for (int m = 0; m < (N + 1) * NX; ++m) xt[m] += w.dx[m];
                for (int m = 0; m < N * NU; ++m) ut[m] += w.du[m];
            }
            have = (qs == 0);
            const double dcmd = xt[NX + 4], tcmd = xt[NX + 5];
            double x[6] = {wst[0], wst[1], wst[2], wst[3], dcmd, tcmd}, u0[2] = {0, 0}, pp[6] = {0, 0, 0, 0, pv[i * 2], 0};
            double phi[6], A[36], Bm[12];
            orc_rk4_sens(&pc, x, u0, pp, phi, A, Bm);
            wst[0] = phi[0]; wst[1] = phi[1]; wst[2] = phi[2]; wst[3] = phi[3]; wst[4] = dcmd; wst[5] = tcmd;
            stats[(size_t)s * Bn + i] = st;
        }
        free(xt); free(wsd);
    }
}
