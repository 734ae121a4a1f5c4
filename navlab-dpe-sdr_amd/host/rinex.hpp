// rinex.hpp -- RINEX 2.x GPS navigation-message reader + the reference's ephemeris choice, for dsp::DPInit.
//
// The reference's DPInit opens a RINEX nav file next to the handoff CSV (cudarecv/modules/src/dpinit.cpp:130-144,
// parser adapted from RTKLIB in cudarecv/utils/src/rinexparse.cpp), groups the records into sets keyed by the integer
// toe in order of first appearance (ReadRinexBody :186-222; a later record of the same (toe, PRN) replaces the earlier
// one) and cuChanMgr picks per PRN the first set that holds it unless a later set's toe is STRICTLY closer to the
// transmit time (cuchanmgr.cu:269-299; seconds of week only).  Same rules here; Python twin: rinex.py.
// Record layout (RINEX 2.10 nav): 8 lines; line 1 = PRN I2, toc yy mm dd hh mm ss.s, af0 af1 af2 (D19.12);
// lines 2-8 = 3X, 4 x D19.12.
#pragma once
#include <cmath>
#include <cstdlib>
#include <fstream>
#include <string>
#include <vector>

namespace dsp { namespace utils {

struct RinexNavRecord {
    int prn = 0, week = 0;
    double tocs = 0;      // toc as GPS seconds of week (time2gpst)
    double data[31] = {}; // af0 af1 af2, then 7 lines x 4 fields
    // the 21 values the channel manager takes, in the handoff CSV's order (sqrt_A, e, i_0, OMEGA_0, omega, M_0, delta_n,
    // OMEGADOT, IDOT, C_rc, C_rs, C_uc, C_us, C_ic, C_is, t_oe, t_oc, a_f0, a_f1, a_f2, T_GD)
    void eph21(double *o) const
    {
        static const int idx[21] = {10, 8, 15, 13, 17, 6, 5, 18, 19, 16, 4, 7, 9, 12, 14, 11, -1, 0, 1, 2, 25};   // rinexparse.cpp:324-345
        for (int j = 0; j < 21; ++j) o[j] = idx[j] < 0 ? tocs : data[idx[j]];
    }
    int toe_key() const { return (int)data[11]; }
};

inline double rinex_num(const std::string &line, size_t pos, size_t len)
{
    if (pos >= line.size()) return 0.0;
    std::string s = line.substr(pos, len);
    for (char &c : s)
        if (c == 'D' || c == 'd') c = 'E';
    char *end = nullptr;
    const double v = std::strtod(s.c_str(), &end);
    return end == s.c_str() ? 0.0 : v;
}

// days from 1980-01-06 to (y, m, d), proleptic Gregorian
inline long long days_from_gps_epoch(int y, int m, int d)
{
    auto days_from_civil = [](int yy, int mm, int dd) {
        yy -= mm <= 2;
        const long long era = (yy >= 0 ? yy : yy - 399) / 400;
        const unsigned yoe = (unsigned)(yy - era * 400);
        const unsigned doy = (153 * (mm + (mm > 2 ? -3 : 9)) + 2) / 5 + dd - 1;
        const unsigned doe = yoe * 365 + yoe / 4 - yoe / 100 + doy;
        return era * 146097 + (long long)doe - 719468;
    };
    return days_from_civil(y, m, d) - days_from_civil(1980, 1, 6);
}

// 0 ok, -1 failure (message in err)
inline int read_rinex_nav(const std::string &path, std::vector<RinexNavRecord> &out, std::string &err)
{
    std::ifstream f(path);
    if (!f) { err = "Open RINEXParamsFile failed: " + path; return -1; }
    std::string line;
    double version = 2.10;
    bool header = false;
    while (std::getline(f, line)) {
        const std::string label = line.size() > 60 ? line.substr(60) : "";
        if (label.find("RINEX VERSION / TYPE") != std::string::npos) {
            version = rinex_num(line, 0, 9);
            if (line.size() > 20 && line[20] != 'N') { err = "Unsupported RINEX body type"; return -1; }
        }
        if (label.find("END OF HEADER") != std::string::npos) { header = true; break; }
    }
    if (!header) { err = "Failed to read RINEX header"; return -1; }
    if (version >= 3.0) { err = "Received unsupported RINEX version (should be < 3.0)"; return -1; }
    std::vector<std::string> rec;
    while (std::getline(f, line)) {
        while (!line.empty() && (line.back() == '\r' || line.back() == '\n')) line.pop_back();
        if (rec.empty() && line.find_first_not_of(' ') == std::string::npos) continue;
        rec.push_back(line);
        if (rec.size() < 8) continue;
        RinexNavRecord r;
        const std::string &l0 = rec[0];
        r.prn = (int)rinex_num(l0, 0, 2);
        const int yy = (int)rinex_num(l0, 3, 2), mo = (int)rinex_num(l0, 6, 2), dd = (int)rinex_num(l0, 9, 2);
        const int hh = (int)rinex_num(l0, 12, 2), mi = (int)rinex_num(l0, 15, 2);
        const double ss = rinex_num(l0, 17, 5);
        const long long days = days_from_gps_epoch(yy + (yy < 80 ? 2000 : 1900), mo, dd);
        const long long whole = days * 86400 + hh * 3600 + mi * 60 + (long long)ss;
        r.week = (int)(whole / 604800);
        r.tocs = (double)(whole % 604800) + (ss - std::floor(ss));
        for (int j = 0; j < 3; ++j) r.data[j] = rinex_num(l0, 22 + 19 * j, 19);
        for (int k = 1; k < 8; ++k)
            for (int j = 0; j < 4; ++j) r.data[3 + 4 * (k - 1) + j] = rinex_num(rec[k], 3 + 19 * j, 19);
        out.push_back(r);
        rec.clear();
    }
    if (out.empty()) { err = "Failed to read in RINEX file"; return -1; }
    return 0;
}

// eph: [nPrn][21]; txTime: seconds of week.  0 ok, -1 when a PRN has no ephemeris.
inline int select_ephemerides(const std::vector<RinexNavRecord> &nav, const std::vector<int> &prns, double txTime,
                              std::vector<double> &eph, std::string &err)
{
    std::vector<int> keys;   // sets in order of first appearance
    for (const RinexNavRecord &r : nav) {
        bool seen = false;
        for (int k : keys) seen = seen || k == r.toe_key();
        if (!seen) keys.push_back(r.toe_key());
    }
    eph.assign(prns.size() * 21, 0.0);
    for (size_t i = 0; i < prns.size(); ++i) {
        int best = -1;
        bool have = false;
        for (int key : keys) {
            bool holds = false;
            for (const RinexNavRecord &r : nav) holds = holds || (r.prn == prns[i] && r.toe_key() == key);
            if (!holds) continue;
            if (!have || std::fabs(key - txTime) < std::fabs(best - txTime)) { best = key; have = true; }
        }
        if (!have) { err = "no ephemeris for PRN " + std::to_string(prns[i]) + " in the RINEX data"; return -1; }
        const RinexNavRecord *pick = nullptr;
        for (const RinexNavRecord &r : nav)
            if (r.prn == prns[i] && r.toe_key() == best) pick = &r;   // the later record of a (toe, PRN) wins
        pick->eph21(&eph[i * 21]);
    }
    return 0;
}

}}  // namespace dsp::utils
