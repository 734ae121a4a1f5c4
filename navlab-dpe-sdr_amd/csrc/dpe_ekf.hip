// dpe_ekf.hip -- the measurement -> state step of the flow: 8-state Kalman filter on the
// host in fp64.  Restates dsp::cuEKF::StepUpdate / StepPredict / GetQVal (cudarecv/modules/src/cuekf.cu:626-742,
// EKF_Update_Q :42-78, F from EKF_MakeDPERandomWalkFMatrix :111-143) and its Python twin
// (pygnss/pythonreceiver/vector/ekf.py:58-177).  The reference runs these 8x8 products through cuBLAS
// Dgemm/Dgemv + batched LU on the GPU with two stream synchronisations per step; 8x8 fp64 is a few hundred
// flops, so here it is plain host code (no kernel: a launch would cost more than the arithmetic).
// The shipped flow disables the filter (dpeflow.cpp:90 -> EKF_PassMeas); this is the EnableEKF=true path.
// Matrices are ROW-major at this boundary.
#include <cmath>
#include <cstring>

#include "dpe_common.h"

struct dpe_ekf {
    double F[64], H[64], Q[64], K[64];
    double Pk1k1[64], Pkk1[64];
    double xk1k1[8], xkk1[8];
    double lpfVals[20];   // running average of |velocity| over 20 updates (EKF_Update_Q :52-57)
    int lpfIdx;
    double lpfAvg;
};

namespace {

constexpr int N = 8;

void eye(double *a)
{
    for (int i = 0; i < N * N; ++i) a[i] = 0.0;
    for (int i = 0; i < N; ++i) a[i * N + i] = 1.0;
}

// c = a * b  (or a * b^T)
void mul(const double *a, const double *b, bool bT, double *c)
{
    double t[64];
    for (int i = 0; i < N; ++i)
        for (int j = 0; j < N; ++j) {
            double s = 0.0;
            for (int k = 0; k < N; ++k) s += a[i * N + k] * (bT ? b[j * N + k] : b[k * N + j]);
            t[i * N + j] = s;
        }
    memcpy(c, t, sizeof(t));
}

// LU with partial pivoting, then the inverse column by column (getrf + getri, cuekf.cu:681-694)
bool invert(const double *a, double *inv)
{
    double lu[64];
    int piv[N];
    memcpy(lu, a, sizeof(lu));
    for (int i = 0; i < N; ++i) piv[i] = i;
    for (int c = 0; c < N; ++c) {
        int p = c;
        for (int r = c + 1; r < N; ++r)
            if (std::fabs(lu[r * N + c]) > std::fabs(lu[p * N + c])) p = r;
        if (lu[p * N + c] == 0.0) return false;
        if (p != c) {
            for (int k = 0; k < N; ++k) std::swap(lu[p * N + k], lu[c * N + k]);
            std::swap(piv[p], piv[c]);
        }
        for (int r = c + 1; r < N; ++r) {
            lu[r * N + c] /= lu[c * N + c];
            for (int k = c + 1; k < N; ++k) lu[r * N + k] -= lu[r * N + c] * lu[c * N + k];
        }
    }
    for (int col = 0; col < N; ++col) {
        double y[N];
        for (int i = 0; i < N; ++i) {
            double s = (piv[i] == col) ? 1.0 : 0.0;
            for (int k = 0; k < i; ++k) s -= lu[i * N + k] * y[k];
            y[i] = s;
        }
        for (int i = N - 1; i >= 0; --i) {
            double s = y[i];
            for (int k = i + 1; k < N; ++k) s -= lu[i * N + k] * inv[k * N + col];
            inv[i * N + col] = s / lu[i * N + i];
        }
    }
    return true;
}

}  // namespace

extern "C" {

int dpe_ekf_create(const dpe_ekf_config *cfg, dpe_ekf **out)
{
    DPE_REQUIRE(cfg && out, "[cuEKF] create: null argument");
    DPE_REQUIRE(cfg->sampleLength > 0, "[cuEKF] create: SampleLength must be positive");
    dpe_ekf *h = new dpe_ekf();
    eye(h->F);
    if (cfg->coupleVelocity)
        for (int j = 0; j < 4; ++j) h->F[j * N + j + 4] = cfg->sampleLength;   // EKF_MakeDPERandomWalkFMatrix :133-139
    eye(h->H);                                                                // :461
    eye(h->Q);                                                                // :463
    eye(h->K);
    eye(h->Pkk1);                                                             // :464
    memcpy(h->Pk1k1, cfg->P0, sizeof(h->Pk1k1));                              // InitP, :352
    memcpy(h->xk1k1, cfg->x0, sizeof(h->xk1k1));                              // InitX, :338-344
    memcpy(h->xkk1, cfg->x0, sizeof(h->xkk1));
    for (double &v : h->lpfVals) v = 0.0;                                     // :475-477
    h->lpfIdx = 0;
    h->lpfAvg = 0.0;
    *out = h;
    return 0;
}

int dpe_ekf_destroy(dpe_ekf *h)
{
    delete h;
    return 0;
}

// k|k from k|k-1 (StepUpdate, cuekf.cu:660-721)
int dpe_ekf_step_update(dpe_ekf *h, const double *z, const double *R)
{
    DPE_REQUIRE(h && z && R, "[cuEKF] StepUpdate: null argument");
    double y[N], S[64], Sinv[64], T[64];
    for (int i = 0; i < N; ++i) {                                             // y = z - H x_k|k-1
        double s = z[i];
        for (int k = 0; k < N; ++k) s -= h->H[i * N + k] * h->xkk1[k];
        y[i] = s;
    }
    mul(h->H, h->Pkk1, false, T);                                             // S = H P H^T + R
    mul(T, h->H, true, S);
    for (int i = 0; i < 64; ++i) S[i] += R[i];
    if (!invert(S, Sinv)) {
        dpe::set_error("[cuEKF] Error: StepUpdate() S inversion failed");    // :683-687
        return -1;
    }
    mul(h->Pkk1, h->H, true, T);                                              // K = P H^T S^-1
    mul(T, Sinv, false, h->K);
    for (int i = 0; i < N; ++i) {                                             // x_k|k = x_k|k-1 + K y
        double s = h->xkk1[i];
        for (int k = 0; k < N; ++k) s += h->K[i * N + k] * y[k];
        h->xk1k1[i] = s;
    }
    double IKH[64];                                                           // P_k|k = (I - K H) P_k|k-1
    mul(h->K, h->H, false, IKH);
    for (int i = 0; i < 64; ++i) IKH[i] = -IKH[i];
    for (int i = 0; i < N; ++i) IKH[i * N + i] += 1.0;
    mul(IKH, h->Pkk1, false, h->Pk1k1);
    return 0;
}

// k+1|k from k|k (StepPredict :626-656 with GetQVal :733-742 and EKF_Update_Q :42-78)
int dpe_ekf_step_predict(dpe_ekf *h)
{
    DPE_REQUIRE(h, "[cuEKF] StepPredict: null handle");
    const double *x = h->xk1k1;
    const double v = std::sqrt(x[4] * x[4] + x[5] * x[5] + x[6] * x[6]);
    h->lpfAvg = h->lpfAvg - h->lpfVals[h->lpfIdx] + (v / 20.0);
    h->lpfVals[h->lpfIdx] = v / 20.0;
    if (++h->lpfIdx >= 20) h->lpfIdx = 0;
    const double q = 1.0 + 250.0 / std::fmin(std::fmax(h->lpfAvg * h->lpfAvg, 50.0), 125.0);
    double Q0[64], T[64];
    for (double &e : Q0) e = 0.0;
    Q0[4 * N + 4] = Q0[5 * N + 5] = Q0[6 * N + 6] = q;
    Q0[7 * N + 7] = (2.5e-10) * (2.5e-10) * dpe::kC * dpe::kC;                 // Q_CLOCK_DRIFT, cuekf.h:28
    mul(h->F, Q0, false, T);                                                  // Q = F Q F^T
    mul(T, h->F, true, h->Q);
    for (int i = 0; i < N; ++i) {                                             // x_k|k-1 = F x_k-1|k-1
        double s = 0.0;
        for (int k = 0; k < N; ++k) s += h->F[i * N + k] * h->xk1k1[k];
        h->xkk1[i] = s;
    }
    mul(h->F, h->Pk1k1, false, T);                                            // P_k|k-1 = F P F^T + Q
    mul(T, h->F, true, h->Pkk1);
    for (int i = 0; i < 64; ++i) h->Pkk1[i] += h->Q[i];
    return 0;
}

int dpe_ekf_state(dpe_ekf *h, double *xk1k1, double *xkk1, double *Pk1k1, double *Pkk1, double *Q, double *K)
{
    DPE_REQUIRE(h, "[cuEKF] state: null handle");
    if (xk1k1) memcpy(xk1k1, h->xk1k1, sizeof(h->xk1k1));
    if (xkk1) memcpy(xkk1, h->xkk1, sizeof(h->xkk1));
    if (Pk1k1) memcpy(Pk1k1, h->Pk1k1, sizeof(h->Pk1k1));
    if (Pkk1) memcpy(Pkk1, h->Pkk1, sizeof(h->Pkk1));
    if (Q) memcpy(Q, h->Q, sizeof(h->Q));
    if (K) memcpy(K, h->K, sizeof(h->K));
    return 0;
}

}  // extern "C"
