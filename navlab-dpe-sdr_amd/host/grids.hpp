// grids.hpp -- manifold grid builders and the rngrid CSV loader (host side of BatchCorrManifold::Start).
#pragma once
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

namespace dsp { namespace utils {

enum ManifoldGridTypes { Uniform, Exponential, ArthurBasis };  // cudarecv/utils/inc/gridhelper.h:30-35

// One axis of BCM_InitPosGrid (batchcorrmanifold.cu:176-246): index -> offset from the centre.
inline double grid_axis(ManifoldGridTypes type, int idx, int dim, double spacing)
{
    const int half = (dim - 1) / 2;   // gridHalfIdx, :2331
    if (type == ArthurBasis && (idx < half / 2 || (dim - idx) < half / 2)) {
        // outer quarters: three times the spacing, shifted so the axis stays continuous (:192-199)
        const double shift = spacing * ((half / 2) + 1) * 2;
        return 3 * spacing * (idx - half) + (idx < half ? shift : -shift);
    }
    return spacing * (idx - half);
}

// Tensor grid, x slowest / t fastest (:165-170).  out: G x 4 {x,y,z,delta_t}; timeGrid: dim[3] entries.
inline void build_grid(ManifoldGridTypes type, const int dim[4], const double spacing[4], std::vector<double> &out,
                       std::vector<double> *timeGrid, bool velocity)
{
    const long long G = (long long)dim[0] * dim[1] * dim[2] * dim[3];
    out.resize((size_t)G * 4);
    // the reference's velocity grid is uniform for every grid type (BCM_InitVelGrid :293-307)
    const ManifoldGridTypes t = velocity ? Uniform : type;
    for (long long i = 0; i < G; ++i) {
        long long r = i;
        const int it = (int)(r % dim[3]); r /= dim[3];
        const int iz = (int)(r % dim[2]); r /= dim[2];
        const int iy = (int)(r % dim[1]);
        const int ix = (int)(r / dim[1]);
        double *p = &out[(size_t)i * 4];
        p[0] = grid_axis(t, ix, dim[0], spacing[0]);
        p[1] = grid_axis(t, iy, dim[1], spacing[1]);
        p[2] = grid_axis(t, iz, dim[2], spacing[2]);
        p[3] = grid_axis(t, it, dim[3], spacing[3]);
    }
    if (timeGrid) {
        timeGrid->resize(dim[3]);
        for (int it = 0; it < dim[3]; ++it) (*timeGrid)[it] = grid_axis(t, it, dim[3], spacing[3]);
    }
}

// "x,y,z,delta_t\r\n" per line (batchcorrmanifold.cu:2433-2444).  Unlike the reference the row count is
// checked against `expect` (the reference overruns its buffer when they differ).
inline int load_grid_csv(const std::string &path, long long expect, std::vector<double> &out)
{
    FILE *f = std::fopen(path.c_str(), "r");
    if (!f) return -1;
    out.clear();
    char line[1024];
    while (std::fgets(line, sizeof(line), f)) {
        double v[4];
        char *p = line;
        int n = 0;
        for (; n < 4; ++n) {
            char *end = nullptr;
            v[n] = std::strtod(p, &end);
            if (end == p) break;
            p = end;
            while (*p == ',' || *p == ' ') ++p;
        }
        if (n == 0) continue;   // blank line
        if (n != 4) { std::fclose(f); return -2; }
        out.insert(out.end(), v, v + 4);
    }
    std::fclose(f);
    if (expect > 0 && (long long)out.size() != expect * 4) return -3;
    return 0;
}

// Bank half-widths that cover every index a grid can reach.
inline void bank_half_widths(const std::vector<double> &pos, const std::vector<double> &vel, double fs, long long nfft,
                             int *lagHalf, int *binHalf)
{
    double ep = 0, ev = 0;
    for (size_t i = 0; i + 3 < pos.size(); i += 4) {
        const double e = std::sqrt(pos[i] * pos[i] + pos[i + 1] * pos[i + 1] + pos[i + 2] * pos[i + 2]) + std::fabs(pos[i + 3]);
        if (e > ep) ep = e;
    }
    for (size_t i = 0; i + 3 < vel.size(); i += 4) {
        const double e = std::sqrt(vel[i] * vel[i] + vel[i + 1] * vel[i + 1] + vel[i + 2] * vel[i + 2]) + std::fabs(vel[i + 3]);
        if (e > ev) ev = e;
    }
    *lagHalf = (int)std::ceil(ep * fs / 299792458.0) + 2;
    *binHalf = (int)std::ceil(ev * ((double)nfft / fs) * 1.57542e9 / 299792458.0) + 3;
}

}}  // namespace dsp::utils
