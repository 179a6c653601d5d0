// orbit.cpp — see orbit.h.  NORAD SGP4 and SDP4 (Spacetrack Report #3: near-earth and deep-space models), WGS-72.
#include "orbit.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <fstream>
#include <vector>

namespace dpx {

namespace {

constexpr double kPi = 3.14159265358979323846;
constexpr double kTwoPi = 2.0 * kPi;
constexpr double kDeg = kPi / 180.0;
constexpr double kAe = 1.0;
constexpr double kTothrd = 2.0 / 3.0;
constexpr double kXkmper = 6378.135;            // WGS-72 earth radius, km
constexpr double kF = 1.0 / 298.26;             // WGS-72 flattening
constexpr double kGe = 398600.8;                // km^3 / s^2
constexpr double kJ2 = 1.0826158e-3, kJ3 = -2.53881e-6, kJ4 = -1.65597e-6;
constexpr double kCk2 = kJ2 / 2.0, kCk4 = -3.0 * kJ4 / 8.0, kXj3 = kJ3;
constexpr double kQo = kAe + 120.0 / kXkmper, kS = kAe + 78.0 / kXkmper;
constexpr double kSecDay = 86400.0;
constexpr double kOmegaE = 1.00273790934;       // earth rotations per sidereal day

double xke() { return sqrt(3600.0 * kGe / (kXkmper * kXkmper * kXkmper)); }
double qoms2t() { const double d = kQo - kS; return d * d * d * d; }

double fmod2p(double x)
{
    x = fmod(x, kTwoPi);
    return x < 0 ? x + kTwoPi : x;
}

// "  12345-3" style field with implied leading decimal point
bool implied_exp(const std::string &f, double *out)
{
    std::string s;
    for (char c : f) if (c != ' ') s.push_back(c);
    if (s.empty()) { *out = 0; return true; }
    size_t i = 0;
    double sign = 1;
    if (s[i] == '-') { sign = -1; ++i; } else if (s[i] == '+') ++i;
    size_t e = s.find_last_of("+-");
    std::string mant = (e != std::string::npos && e > i) ? s.substr(i, e - i) : s.substr(i);
    int ex = (e != std::string::npos && e > i) ? atoi(s.substr(e).c_str()) : 0;
    for (char c : mant) if (c < '0' || c > '9') return false;
    *out = sign * atof(("0." + mant).c_str()) * pow(10.0, ex);
    return true;
}

double jd_of_year(int year)   // Julian date of Jan 0.0 UTC of `year`
{
    const int y = year - 1;
    const int a = y / 100, b = 2 - a + a / 4;
    return floor(365.25 * y) + floor(30.6001 * 14) + 1720994.5 + b;
}

double theta_g(double jd)     // Greenwich mean sidereal angle, rad
{
    double ut = jd + 0.5;
    ut -= floor(ut);
    const double jd0 = jd - ut;
    const double tu = (jd0 - 2451545.0) / 36525.0;
    double gmst = 24110.54841 + tu * (8640184.812866 + tu * (0.093104 - tu * 6.2e-6));
    gmst = fmod(gmst + kSecDay * kOmegaE * ut, kSecDay);
    if (gmst < 0) gmst += kSecDay;
    return kTwoPi * gmst / kSecDay;
}

}  // namespace

double unix_to_jd(double t) { return 2440587.5 + t / kSecDay; }

bool parse_utc(const char *s, int64_t *out)
{
    int Y, M, D, h, m, sec;
    char tail;
    if (sscanf(s, "%d-%d-%dT%d:%d:%d%c", &Y, &M, &D, &h, &m, &sec, &tail) != 6) return false;
    if (M < 1 || M > 12 || D < 1 || D > 31 || h < 0 || h > 23 || m < 0 || m > 59 || sec < 0 || sec > 60) return false;
    // days from civil (proleptic Gregorian)
    int y = Y - (M <= 2);
    const int era = (y >= 0 ? y : y - 399) / 400;
    const unsigned yoe = (unsigned)(y - era * 400);
    const unsigned doy = (153u * (unsigned)(M + (M > 2 ? -3 : 9)) + 2) / 5 + (unsigned)D - 1;
    const unsigned doe = yoe * 365 + yoe / 4 - yoe / 100 + doy;
    const int64_t days = (int64_t)era * 146097 + (int64_t)doe - 719468;
    *out = days * 86400 + h * 3600 + m * 60 + sec;
    return true;
}

bool tle_parse(const char *l1, const char *l2, Tle *out, std::string *err)
{
    const std::string a(l1), b(l2);
    if (a.size() < 64 || b.size() < 63 || a[0] != '1' || b[0] != '2') {
        if (err) *err = "malformed two-line element set";
        return false;
    }
    const int yy = atoi(a.substr(18, 2).c_str());
    const double day = atof(a.substr(20, 12).c_str());
    const int year = yy < 57 ? 2000 + yy : 1900 + yy;
    out->epoch_jd = jd_of_year(year) + day;
    if (!implied_exp(a.substr(53, 8), &out->bstar)) {
        if (err) *err = "malformed B* field";
        return false;
    }
    out->xincl = atof(b.substr(8, 8).c_str()) * kDeg;
    out->xnodeo = atof(b.substr(17, 8).c_str()) * kDeg;
    out->eo = atof(("0." + b.substr(26, 7)).c_str());
    out->omegao = atof(b.substr(34, 8).c_str()) * kDeg;
    out->xmo = atof(b.substr(43, 8).c_str()) * kDeg;
    out->xno = atof(b.substr(52, 11).c_str()) * kTwoPi / 1440.0;
    if (!(out->xno > 0) || !(out->eo >= 0 && out->eo < 1)) {
        if (err) *err = "element set out of range";
        return false;
    }
    return true;
}

bool tle_from_file(const char *path, const char *name, Tle *out, std::string *err)
{
    std::ifstream f(path);
    if (!f) {
        if (err) *err = std::string("cannot open TLE file ") + path;
        return false;
    }
    auto trim = [](std::string s) {
        while (!s.empty() && (s.back() == '\r' || s.back() == '\n' || s.back() == ' ' || s.back() == '\t')) s.pop_back();
        size_t i = 0;
        while (i < s.size() && (s[i] == ' ' || s[i] == '\t')) ++i;
        return s.substr(i);
    };
    const std::string want = trim(name);
    std::vector<std::string> lines;
    std::string ln;
    while (std::getline(f, ln)) lines.push_back(ln);
    for (size_t i = 0; i + 2 < lines.size(); ++i) {
        if (trim(lines[i]) == want && lines[i + 1].size() > 0 && lines[i + 1][0] == '1' && lines[i + 2].size() > 0 &&
            lines[i + 2][0] == '2') {
            out->name = want;
            return tle_parse(lines[i + 1].c_str(), lines[i + 2].c_str(), out, err);
        }
    }
    if (err) *err = "TLE '" + want + "' not found in " + path;
    return false;
}

bool Sgp4::init(const Tle &t, std::string *err)
{
    tle_ = t;
    const double ke = xke();
    const double a1 = pow(ke / t.xno, kTothrd);
    cosio = cos(t.xincl);
    sinio = sin(t.xincl);
    const double theta2 = cosio * cosio;
    x3thm1 = 3.0 * theta2 - 1.0;
    const double eosq = t.eo * t.eo, betao2 = 1.0 - eosq, betao = sqrt(betao2);
    const double del1 = 1.5 * kCk2 * x3thm1 / (a1 * a1 * betao * betao2);
    const double ao = a1 * (1.0 - del1 * (0.5 * kTothrd + del1 * (1.0 + 134.0 / 81.0 * del1)));
    const double delo = 1.5 * kCk2 * x3thm1 / (ao * ao * betao * betao2);
    xnodp = t.xno / (1.0 + delo);
    aodp = ao / (1.0 - delo);
    (void)err;
    deep_ = kTwoPi / xnodp >= 225.0;               // Spacetrack Report #3: periods of 225 minutes and more take the deep-space model
    simple_ = !deep_ && (aodp * (1.0 - t.eo) / kAe) < (220.0 / kXkmper + kAe);
    double s4 = kS, qoms24 = qoms2t();
    const double perige = (aodp * (1.0 - t.eo) - kAe) * kXkmper;
    if (perige < 156.0) {
        s4 = perige <= 98.0 ? 20.0 : perige - 78.0;
        qoms24 = pow((120.0 - s4) * kAe / kXkmper, 4.0);
        s4 = s4 / kXkmper + kAe;
    }
    const double pinvsq = 1.0 / (aodp * aodp * betao2 * betao2);
    const double tsi = 1.0 / (aodp - s4);
    eta = aodp * t.eo * tsi;
    const double etasq = eta * eta, eeta = t.eo * eta, psisq = fabs(1.0 - etasq);
    const double coef = qoms24 * pow(tsi, 4.0), coef1 = coef / pow(psisq, 3.5);
    const double c2 = coef1 * xnodp * (aodp * (1.0 + 1.5 * etasq + eeta * (4.0 + etasq)) +
                                       0.75 * kCk2 * tsi / psisq * x3thm1 * (8.0 + 3.0 * etasq * (8.0 + etasq)));
    c1 = t.bstar * c2;
    const double a3ovk2 = -kXj3 / kCk2 * kAe * kAe * kAe;
    const double c3 = t.eo > 1e-12 ? coef * tsi * a3ovk2 * xnodp * kAe * sinio / t.eo : 0.0;
    x1mth2 = 1.0 - theta2;
    c4 = 2.0 * xnodp * coef1 * aodp * betao2 *
         (eta * (2.0 + 0.5 * etasq) + t.eo * (0.5 + 2.0 * etasq) -
          2.0 * kCk2 * tsi / (aodp * psisq) *
              (-3.0 * x3thm1 * (1.0 - 2.0 * eeta + etasq * (1.5 - 0.5 * eeta)) +
               0.75 * x1mth2 * (2.0 * etasq - eeta * (1.0 + etasq)) * cos(2.0 * t.omegao)));
    c5 = 2.0 * coef1 * aodp * betao2 * (1.0 + 2.75 * (etasq + eeta) + eeta * etasq);
    const double theta4 = theta2 * theta2;
    const double temp1 = 3.0 * kCk2 * pinvsq * xnodp, temp2 = temp1 * kCk2 * pinvsq;
    const double temp3 = 1.25 * kCk4 * pinvsq * pinvsq * xnodp;
    xmdot = xnodp + 0.5 * temp1 * betao * x3thm1 + 0.0625 * temp2 * betao * (13.0 - 78.0 * theta2 + 137.0 * theta4);
    const double x1m5th = 1.0 - 5.0 * theta2;
    omgdot = -0.5 * temp1 * x1m5th + 0.0625 * temp2 * (7.0 - 114.0 * theta2 + 395.0 * theta4) +
             temp3 * (3.0 - 36.0 * theta2 + 49.0 * theta4);
    const double xhdot1 = -temp1 * cosio;
    xnodot = xhdot1 + (0.5 * temp2 * (4.0 - 19.0 * theta2) + 2.0 * temp3 * (3.0 - 7.0 * theta2)) * cosio;
    omgcof = t.bstar * c3 * cos(t.omegao);
    xmcof = fabs(eeta) > 1e-12 ? -kTothrd * coef * t.bstar * kAe / eeta : 0.0;
    xnodcf = 3.5 * betao2 * xhdot1 * c1;
    t2cof = 1.5 * c1;
    xlcof = 0.125 * a3ovk2 * sinio * (3.0 + 5.0 * cosio) / (1.0 + cosio);
    aycof = 0.25 * a3ovk2 * sinio;
    delmo = pow(1.0 + eta * cos(t.xmo), 3.0);
    sinmo = sin(t.xmo);
    x7thm1 = 7.0 * theta2 - 1.0;
    if (deep_) {
        deep_init();
        return true;
    }
    if (!simple_) {
        const double c1sq = c1 * c1;
        d2 = 4.0 * aodp * tsi * c1sq;
        const double temp = d2 * tsi * c1 / 3.0;
        d3 = (17.0 * aodp + s4) * temp;
        d4 = 0.5 * temp * aodp * tsi * (221.0 * aodp + 31.0 * s4) * c1;
        t3cof = d2 + 2.0 * c1sq;
        t4cof = 0.25 * (3.0 * d3 + c1 * (12.0 * d2 + 10.0 * c1sq));
        t5cof = 0.2 * (3.0 * d4 + 12.0 * c1 * d3 + 6.0 * d2 * d2 + 15.0 * c1sq * (2.0 * d2 + c1sq));
    }
    return true;
}

void Sgp4::propagate(double ts, double pos[3], double vel[3]) const
{
    if (deep_) {
        propagate_deep(ts, pos, vel);
        return;
    }
    const Tle &t = tle_;
    const double ke = xke();
    const double xmdf = t.xmo + xmdot * ts, omgadf = t.omegao + omgdot * ts, xnoddf = t.xnodeo + xnodot * ts;
    double omega = omgadf, xmp = xmdf;
    const double tsq = ts * ts;
    const double xnode = xnoddf + xnodcf * tsq;
    double tempa = 1.0 - c1 * ts, tempe = t.bstar * c4 * ts, templ = t2cof * tsq;
    if (!simple_) {
        const double delomg = omgcof * ts;
        const double delm = xmcof * (pow(1.0 + eta * cos(xmdf), 3.0) - delmo);
        const double temp = delomg + delm;
        xmp = xmdf + temp;
        omega = omgadf - temp;
        const double tcube = tsq * ts, tfour = ts * tcube;
        tempa = tempa - d2 * tsq - d3 * tcube - d4 * tfour;
        tempe = tempe + t.bstar * c5 * (sin(xmp) - sinmo);
        templ = templ + t3cof * tcube + tfour * (t4cof + ts * t5cof);
    }
    const double a = aodp * tempa * tempa;
    const double e = t.eo - tempe;
    const double xl = xmp + omega + xnode + xnodp * templ;
    const double beta = sqrt(1.0 - e * e);
    const double xn = ke / pow(a, 1.5);
    // long-period periodics
    const double axn = e * cos(omega);
    double temp = 1.0 / (a * beta * beta);
    const double xll = temp * xlcof * axn, aynl = temp * aycof;
    const double xlt = xl + xll, ayn = e * sin(omega) + aynl;
    // Kepler's equation
    const double capu = fmod2p(xlt - xnode);
    double temp2 = capu, sinepw = 0, cosepw = 0, temp3 = 0, temp4 = 0, temp5 = 0, temp6 = 0;
    for (int i = 0; i < 10; ++i) {
        sinepw = sin(temp2);
        cosepw = cos(temp2);
        temp3 = axn * sinepw;
        temp4 = ayn * cosepw;
        temp5 = axn * cosepw;
        temp6 = ayn * sinepw;
        const double epw = (capu - temp4 + temp3 - temp2) / (1.0 - temp5 - temp6) + temp2;
        if (fabs(epw - temp2) <= 1e-6) break;
        temp2 = epw;
    }
    // short-period preliminary quantities
    const double ecose = temp5 + temp6, esine = temp3 - temp4, elsq = axn * axn + ayn * ayn;
    temp = 1.0 - elsq;
    const double pl = a * temp, r = a * (1.0 - ecose);
    double temp1 = 1.0 / r;
    const double rdot = ke * sqrt(a) * esine * temp1, rfdot = ke * sqrt(pl) * temp1;
    temp2 = a * temp1;
    const double betal = sqrt(temp);
    temp3 = 1.0 / (1.0 + betal);
    const double cosu = temp2 * (cosepw - axn + ayn * esine * temp3);
    const double sinu = temp2 * (sinepw - ayn - axn * esine * temp3);
    const double u = atan2(sinu, cosu);
    const double sin2u = 2.0 * sinu * cosu, cos2u = 2.0 * cosu * cosu - 1.0;
    temp = 1.0 / pl;
    temp1 = kCk2 * temp;
    temp2 = temp1 * temp;
    // short periodics
    const double rk = r * (1.0 - 1.5 * temp2 * betal * x3thm1) + 0.5 * temp1 * x1mth2 * cos2u;
    const double uk = u - 0.25 * temp2 * x7thm1 * sin2u;
    const double xnodek = xnode + 1.5 * temp2 * cosio * sin2u;
    const double xinck = t.xincl + 1.5 * temp2 * cosio * sinio * cos2u;
    const double rdotk = rdot - xn * temp1 * x1mth2 * sin2u;
    const double rfdotk = rfdot + xn * temp1 * (x1mth2 * cos2u + 1.5 * x3thm1);
    // orientation vectors
    const double sinuk = sin(uk), cosuk = cos(uk), sinik = sin(xinck), cosik = cos(xinck);
    const double sinnok = sin(xnodek), cosnok = cos(xnodek);
    const double xmx = -sinnok * cosik, xmy = cosnok * cosik;
    const double ux = xmx * sinuk + cosnok * cosuk, uy = xmy * sinuk + sinnok * cosuk, uz = sinik * sinuk;
    const double vx = xmx * cosuk - cosnok * sinuk, vy = xmy * cosuk - sinnok * sinuk, vz = sinik * cosuk;
    pos[0] = rk * ux * kXkmper;
    pos[1] = rk * uy * kXkmper;
    pos[2] = rk * uz * kXkmper;
    const double vs = kXkmper / 60.0;   // earth radii / min -> km / s
    vel[0] = (rdotk * ux + rfdotk * vx) * vs;
    vel[1] = (rdotk * uy + rfdotk * vy) * vs;
    vel[2] = (rdotk * uz + rfdotk * vz) * vs;
}


// ------------------------------------------------------------------ SDP4: the deep-space subroutine of Spacetrack Report #3
namespace {

constexpr double kZns = 1.19459e-5, kC1ss = 2.9864797e-6, kZes = 0.01675, kZnl = 1.5835218e-4, kC1l = 4.7968065e-7, kZel = 0.05490;
constexpr double kZcosis = 0.91744867, kZsinis = 0.39785416, kZsings = -0.98088458, kZcosgs = 0.1945905;
constexpr double kQ22 = 1.7891679e-6, kQ31 = 2.1460748e-6, kQ33 = 2.2123015e-7;
constexpr double kG22 = 5.7686396, kG32 = 0.95240898, kG44 = 1.8014998, kG52 = 1.0508330, kG54 = 4.4108898;
constexpr double kRoot22 = 1.7891679e-6, kRoot32 = 3.7393792e-7, kRoot44 = 7.3636953e-9, kRoot52 = 1.1428639e-7, kRoot54 = 2.1765803e-9;
constexpr double kThdt = 4.3752691e-3;         // earth rotation, rad / min
constexpr double kStep = 720.0, kStep2 = 259200.0;   // resonance integrator: 720-minute steps (and step^2 / 2)

double actan(double sinx, double cosx)       // the report's four-quadrant arctangent, in [0, 2 pi)
{
    double a = atan2(sinx, cosx);
    return a < 0 ? a + kTwoPi : a;
}

// Greenwich sidereal angle at the epoch, as the report computes it (days since 1950 Jan 0.0 UTC)
double theta_g_epoch(double ds50)
{
    const double ts70 = ds50 - 7305.0;
    const double ds70 = floor(ts70 + 1e-8);
    const double trfac = ts70 - ds70;
    constexpr double c1 = 1.72027916940703639e-2, thgr70 = 1.7321343856509374, fk5r = 5.07551419432269442e-15;
    return fmod2p(thgr70 + c1 * ds70 + (c1 + kTwoPi) * trfac + ts70 * ts70 * fk5r);
}

}  // namespace

void Sgp4::deep_init()
{
    const Tle &t = tle_;
    Deep &d = dp_;
    const double ds50 = t.epoch_jd - 2433281.5;
    d.thgr = theta_g_epoch(ds50);
    const double eq = t.eo, eqsq = eq * eq, bsq = 1.0 - eqsq, rteqsq = sqrt(bsq);
    d.xnq = xnodp;
    const double aqnv = 1.0 / aodp;
    d.xqncl = t.xincl;
    const double xpidot = omgdot + xnodot;
    const double sinq = sin(t.xnodeo), cosq = cos(t.xnodeo);
    const double sinomo = sin(t.omegao), cosomo = cos(t.omegao);
    d.omegaq = t.omegao;
    const double siniq = sinio, cosiq = cosio;
    // ---- lunar and solar geometry at the epoch
    const double day = ds50 + 18261.5;
    const double xnodce = 4.5236020 - 9.2422029e-4 * day;
    const double stem = sin(xnodce), ctem = cos(xnodce);
    const double zcosil = 0.91375164 - 0.03568096 * ctem, zsinil = sqrt(1.0 - zcosil * zcosil);
    const double zsinhl = 0.089683511 * stem / zsinil, zcoshl = sqrt(1.0 - zsinhl * zsinhl);
    const double c = 4.7199672 + 0.22997150 * day, gam = 5.8351514 + 0.0019443680 * day;
    d.zmol = fmod2p(c - gam);
    double zx = 0.39785416 * stem / zsinil;
    const double zy = zcoshl * ctem + 0.91744867 * zsinhl * stem;
    zx = actan(zx, zy);
    zx = gam + zx - xnodce;
    const double zcosgl = cos(zx), zsingl = sin(zx);
    d.zmos = fmod2p(6.2565837 + 0.017201977 * day);
    // ---- solar terms, then lunar terms
    double zcosg = kZcosgs, zsing = kZsings, zcosi = kZcosis, zsini = kZsinis, zcosh = cosq, zsinh = sinq;
    double cc = kC1ss, zn = kZns, ze = kZes;
    const double xnoi = 1.0 / d.xnq;
    double se = 0, si = 0, sl = 0, sgh = 0, sh = 0;
    for (int pass = 0; pass < 2; ++pass) {
        const double a1 = zcosg * zcosh + zsing * zcosi * zsinh;
        const double a3 = -zsing * zcosh + zcosg * zcosi * zsinh;
        const double a7 = -zcosg * zsinh + zsing * zcosi * zcosh;
        const double a8 = zsing * zsini;
        const double a9 = zsing * zsinh + zcosg * zcosi * zcosh;
        const double a10 = zcosg * zsini;
        const double a2 = cosiq * a7 + siniq * a8;
        const double a4 = cosiq * a9 + siniq * a10;
        const double a5 = -siniq * a7 + cosiq * a8;
        const double a6 = -siniq * a9 + cosiq * a10;
        const double x1 = a1 * cosomo + a2 * sinomo;
        const double x2 = a3 * cosomo + a4 * sinomo;
        const double x3 = -a1 * sinomo + a2 * cosomo;
        const double x4 = -a3 * sinomo + a4 * cosomo;
        const double x5 = a5 * sinomo, x6 = a6 * sinomo, x7 = a5 * cosomo, x8 = a6 * cosomo;
        const double z31 = 12.0 * x1 * x1 - 3.0 * x3 * x3;
        const double z32 = 24.0 * x1 * x2 - 6.0 * x3 * x4;
        const double z33 = 12.0 * x2 * x2 - 3.0 * x4 * x4;
        double z1 = 3.0 * (a1 * a1 + a2 * a2) + z31 * eqsq;
        double z2 = 6.0 * (a1 * a3 + a2 * a4) + z32 * eqsq;
        double z3 = 3.0 * (a3 * a3 + a4 * a4) + z33 * eqsq;
        const double z11 = -6.0 * a1 * a5 + eqsq * (-24.0 * x1 * x7 - 6.0 * x3 * x5);
        const double z12 = -6.0 * (a1 * a6 + a3 * a5) + eqsq * (-24.0 * (x2 * x7 + x1 * x8) - 6.0 * (x3 * x6 + x4 * x5));
        const double z13 = -6.0 * a3 * a6 + eqsq * (-24.0 * x2 * x8 - 6.0 * x4 * x6);
        const double z21 = 6.0 * a2 * a5 + eqsq * (24.0 * x1 * x5 - 6.0 * x3 * x7);
        const double z22 = 6.0 * (a4 * a5 + a2 * a6) + eqsq * (24.0 * (x2 * x5 + x1 * x6) - 6.0 * (x4 * x7 + x3 * x8));
        const double z23 = 6.0 * a4 * a6 + eqsq * (24.0 * x2 * x6 - 6.0 * x4 * x8);
        z1 = z1 + z1 + bsq * z31;
        z2 = z2 + z2 + bsq * z32;
        z3 = z3 + z3 + bsq * z33;
        const double s3 = cc * xnoi, s2 = -0.5 * s3 / rteqsq, s4 = s3 * rteqsq, s1 = -15.0 * eq * s4;
        const double s5 = x1 * x3 + x2 * x4, s6 = x2 * x3 + x1 * x4, s7 = x2 * x4 - x1 * x3;
        se = s1 * zn * s5;
        si = s2 * zn * (z11 + z13);
        sl = -zn * s3 * (z1 + z3 - 14.0 - 6.0 * eqsq);
        sgh = s4 * zn * (z31 + z33 - 6.0);
        sh = -zn * s2 * (z21 + z23);
        if (d.xqncl < 5.2359877e-2) sh = 0.0;
        d.ee2 = 2.0 * s1 * s6;
        d.e3 = 2.0 * s1 * s7;
        d.xi2 = 2.0 * s2 * z12;
        d.xi3 = 2.0 * s2 * (z13 - z11);
        d.xl2 = -2.0 * s3 * z2;
        d.xl3 = -2.0 * s3 * (z3 - z1);
        d.xl4 = -2.0 * s3 * (-21.0 - 9.0 * eqsq) * ze;
        d.xgh2 = 2.0 * s4 * z32;
        d.xgh3 = 2.0 * s4 * (z33 - z31);
        d.xgh4 = -18.0 * s4 * ze;
        d.xh2 = -2.0 * s2 * z22;
        d.xh3 = -2.0 * s2 * (z23 - z21);
        if (pass == 1) break;
        // the solar terms are kept, the loop runs once more with the moon's geometry
        d.sse = se;
        d.ssi = si;
        d.ssl = sl;
        d.ssh = sh / siniq;
        d.ssg = sgh - cosiq * d.ssh;
        d.se2 = d.ee2; d.si2 = d.xi2; d.sl2 = d.xl2; d.sgh2 = d.xgh2; d.sh2 = d.xh2;
        d.se3 = d.e3; d.si3 = d.xi3; d.sl3 = d.xl3; d.sgh3 = d.xgh3; d.sh3 = d.xh3;
        d.sl4 = d.xl4; d.sgh4 = d.xgh4;
        zcosg = zcosgl; zsing = zsingl; zcosi = zcosil; zsini = zsinil;
        zcosh = zcoshl * cosq + zsinhl * sinq;
        zsinh = sinq * zcoshl - cosq * zsinhl;
        zn = kZnl; cc = kC1l; ze = kZel;
    }
    d.sse += se;
    d.ssi += si;
    d.ssl += sl;
    d.ssg += sgh - cosiq / siniq * sh;
    d.ssh += sh / siniq;
    // ---- geopotential resonance: 24-hour (synchronous) and 12-hour (Molniya, GPS) orbits
    d.resonant = d.synchronous = false;
    double bfact = 0;
    if (d.xnq < 0.0052359877 && d.xnq > 0.0034906585) {
        d.resonant = d.synchronous = true;
        const double g200 = 1.0 + eqsq * (-2.5 + 0.8125 * eqsq), g310 = 1.0 + 2.0 * eqsq, g300 = 1.0 + eqsq * (-6.0 + 6.60937 * eqsq);
        const double f220 = 0.75 * (1.0 + cosiq) * (1.0 + cosiq);
        const double f311 = 0.9375 * siniq * siniq * (1.0 + 3.0 * cosiq) - 0.75 * (1.0 + cosiq);
        double f330 = 1.0 + cosiq;
        f330 = 1.875 * f330 * f330 * f330;
        d.del1 = 3.0 * d.xnq * d.xnq * aqnv * aqnv;
        d.del2 = 2.0 * d.del1 * f220 * g200 * kQ22;
        d.del3 = 3.0 * d.del1 * f330 * g300 * kQ33 * aqnv;
        d.del1 = d.del1 * f311 * g310 * kQ31 * aqnv;
        d.fasx2 = 0.13130908; d.fasx4 = 2.8843198; d.fasx6 = 0.37448087;
        d.xlamo = t.xmo + t.xnodeo + t.omegao - d.thgr;
        bfact = xmdot + xpidot - kThdt + d.ssl + d.ssg + d.ssh;
    } else if (d.xnq >= 8.26e-3 && d.xnq <= 9.24e-3 && eq >= 0.5) {
        d.resonant = true;
        const double eoc = eq * eqsq;
        const double g201 = -0.306 - (eq - 0.64) * 0.440;
        double g211, g310, g322, g410, g422, g520, g533, g521, g532;
        if (eq <= 0.65) {
            g211 = 3.616 - 13.247 * eq + 16.290 * eqsq;
            g310 = -19.302 + 117.390 * eq - 228.419 * eqsq + 156.591 * eoc;
            g322 = -18.9068 + 109.7927 * eq - 214.6334 * eqsq + 146.5816 * eoc;
            g410 = -41.122 + 242.694 * eq - 471.094 * eqsq + 313.953 * eoc;
            g422 = -146.407 + 841.880 * eq - 1629.014 * eqsq + 1083.435 * eoc;
            g520 = -532.114 + 3017.977 * eq - 5740.0 * eqsq + 3708.276 * eoc;
        } else {
            g211 = -72.099 + 331.819 * eq - 508.738 * eqsq + 266.724 * eoc;
            g310 = -346.844 + 1582.851 * eq - 2415.925 * eqsq + 1246.113 * eoc;
            g322 = -342.585 + 1554.908 * eq - 2366.899 * eqsq + 1215.972 * eoc;
            g410 = -1052.797 + 4758.686 * eq - 7193.992 * eqsq + 3651.957 * eoc;
            g422 = -3581.69 + 16178.11 * eq - 24462.77 * eqsq + 12422.52 * eoc;
            g520 = eq <= 0.715 ? 1464.74 - 4664.75 * eq + 3763.64 * eqsq
                               : -5149.66 + 29936.92 * eq - 54087.36 * eqsq + 31324.56 * eoc;
        }
        if (eq < 0.7) {
            g533 = -919.2277 + 4988.61 * eq - 9064.77 * eqsq + 5542.21 * eoc;
            g521 = -822.71072 + 4568.6173 * eq - 8491.4146 * eqsq + 5337.524 * eoc;
            g532 = -853.666 + 4690.25 * eq - 8624.77 * eqsq + 5341.4 * eoc;
        } else {
            g533 = -37995.78 + 161616.52 * eq - 229838.2 * eqsq + 109377.94 * eoc;
            g521 = -51752.104 + 218913.95 * eq - 309468.16 * eqsq + 146349.42 * eoc;
            g532 = -40023.88 + 170470.89 * eq - 242699.48 * eqsq + 115605.82 * eoc;
        }
        const double sini2 = siniq * siniq, cosq2 = cosiq * cosiq;
        const double f220 = 0.75 * (1.0 + 2.0 * cosiq + cosq2);
        const double f221 = 1.5 * sini2;
        const double f321 = 1.875 * siniq * (1.0 - 2.0 * cosiq - 3.0 * cosq2);
        const double f322 = -1.875 * siniq * (1.0 + 2.0 * cosiq - 3.0 * cosq2);
        const double f441 = 35.0 * sini2 * f220;
        const double f442 = 39.3750 * sini2 * sini2;
        const double f522 = 9.84375 * siniq * (sini2 * (1.0 - 2.0 * cosiq - 5.0 * cosq2) + 0.33333333 * (-2.0 + 4.0 * cosiq + 6.0 * cosq2));
        const double f523 = siniq * (4.92187512 * sini2 * (-2.0 - 4.0 * cosiq + 10.0 * cosq2) + 6.56250012 * (1.0 + 2.0 * cosiq - 3.0 * cosq2));
        const double f542 = 29.53125 * siniq * (2.0 - 8.0 * cosiq + cosq2 * (-12.0 + 8.0 * cosiq + 10.0 * cosq2));
        const double f543 = 29.53125 * siniq * (-2.0 - 8.0 * cosiq + cosq2 * (12.0 + 8.0 * cosiq - 10.0 * cosq2));
        const double xno2 = d.xnq * d.xnq, ainv2 = aqnv * aqnv;
        double temp1 = 3.0 * xno2 * ainv2;
        double temp = temp1 * kRoot22;
        d.d2201 = temp * f220 * g201;
        d.d2211 = temp * f221 * g211;
        temp1 *= aqnv;
        temp = temp1 * kRoot32;
        d.d3210 = temp * f321 * g310;
        d.d3222 = temp * f322 * g322;
        temp1 *= aqnv;
        temp = 2.0 * temp1 * kRoot44;
        d.d4410 = temp * f441 * g410;
        d.d4422 = temp * f442 * g422;
        temp1 *= aqnv;
        temp = temp1 * kRoot52;
        d.d5220 = temp * f522 * g520;
        d.d5232 = temp * f523 * g532;
        temp = 2.0 * temp1 * kRoot54;
        d.d5421 = temp * f542 * g521;
        d.d5433 = temp * f543 * g533;
        d.xlamo = t.xmo + t.xnodeo + t.xnodeo - d.thgr - d.thgr;
        bfact = xmdot + xnodot + xnodot - kThdt - kThdt + d.ssl + d.ssh + d.ssh;
    }
    if (d.resonant) d.xfact = bfact - d.xnq;
}

// secular effects of the moon and the sun, and the resonance integration (the report's entry DPSEC)
void Sgp4::deep_secular(double t, double *xll, double *omgadf, double *xnode, double *em, double *xinc, double *xn) const
{
    const Deep &d = dp_;
    *xll += d.ssl * t;
    *omgadf += d.ssg * t;
    *xnode += d.ssh * t;
    *em = tle_.eo + d.sse * t;
    *xinc = tle_.xincl + d.ssi * t;
    if (*xinc < 0) {
        *xinc = -*xinc;
        *xnode += kPi;
        *omgadf -= kPi;
    }
    if (!d.resonant) return;
    // The report keeps the integrator's state between calls and restarts at the epoch when the time runs the other way; the
    // steps are fixed 720-minute steps from the epoch either way, so integrating from the epoch at every call gives the same state.
    double atime = 0.0, xni = d.xnq, xli = d.xlamo;
    const double delt = t >= 0 ? kStep : -kStep;
    double xndot = 0, xnddt = 0, xldot = 0;
    auto derivatives = [&]() {
        if (d.synchronous) {
            xndot = d.del1 * sin(xli - d.fasx2) + d.del2 * sin(2.0 * (xli - d.fasx4)) + d.del3 * sin(3.0 * (xli - d.fasx6));
            xnddt = d.del1 * cos(xli - d.fasx2) + 2.0 * d.del2 * cos(2.0 * (xli - d.fasx4)) + 3.0 * d.del3 * cos(3.0 * (xli - d.fasx6));
        } else {
            const double xomi = d.omegaq + omgdot * atime, x2omi = xomi + xomi, x2li = xli + xli;
            xndot = d.d2201 * sin(x2omi + xli - kG22) + d.d2211 * sin(xli - kG22) + d.d3210 * sin(xomi + xli - kG32) +
                    d.d3222 * sin(-xomi + xli - kG32) + d.d4410 * sin(x2omi + x2li - kG44) + d.d4422 * sin(x2li - kG44) +
                    d.d5220 * sin(xomi + xli - kG52) + d.d5232 * sin(-xomi + xli - kG52) + d.d5421 * sin(xomi + x2li - kG54) +
                    d.d5433 * sin(-xomi + x2li - kG54);
            xnddt = d.d2201 * cos(x2omi + xli - kG22) + d.d2211 * cos(xli - kG22) + d.d3210 * cos(xomi + xli - kG32) +
                    d.d3222 * cos(-xomi + xli - kG32) + d.d5220 * cos(xomi + xli - kG52) + d.d5232 * cos(-xomi + xli - kG52) +
                    2.0 * (d.d4410 * cos(x2omi + x2li - kG44) + d.d4422 * cos(x2li - kG44) + d.d5421 * cos(xomi + x2li - kG54) +
                           d.d5433 * cos(-xomi + x2li - kG54));
        }
        xldot = xni + d.xfact;
        xnddt *= xldot;
    };
    while (fabs(t - atime) >= kStep) {
        derivatives();
        xli += xldot * delt + xndot * kStep2;
        xni += xndot * delt + xnddt * kStep2;
        atime += delt;
    }
    derivatives();
    const double ft = t - atime;
    *xn = xni + xndot * ft + xnddt * ft * ft * 0.5;
    const double xl = xli + xldot * ft + xndot * ft * ft * 0.5;
    const double temp = -*xnode + d.thgr + t * kThdt;
    *xll = d.synchronous ? xl - *omgadf + temp : xl + temp + temp;
}

// lunar-solar periodics (the report's entry DPPER), evaluated at every call
void Sgp4::deep_periodic(double t, double *em, double *xinc, double *omgadf, double *xnode, double *xll) const
{
    const Deep &d = dp_;
    const double sinis = sin(*xinc), cosis = cos(*xinc);
    double zm = d.zmos + kZns * t;
    double zf = zm + 2.0 * kZes * sin(zm);
    double sinzf = sin(zf), f2 = 0.5 * sinzf * sinzf - 0.25, f3 = -0.5 * sinzf * cos(zf);
    const double ses = d.se2 * f2 + d.se3 * f3, sis = d.si2 * f2 + d.si3 * f3;
    const double sls = d.sl2 * f2 + d.sl3 * f3 + d.sl4 * sinzf;
    const double sghs = d.sgh2 * f2 + d.sgh3 * f3 + d.sgh4 * sinzf, shs = d.sh2 * f2 + d.sh3 * f3;
    zm = d.zmol + kZnl * t;
    zf = zm + 2.0 * kZel * sin(zm);
    sinzf = sin(zf);
    f2 = 0.5 * sinzf * sinzf - 0.25;
    f3 = -0.5 * sinzf * cos(zf);
    const double sel = d.ee2 * f2 + d.e3 * f3, sil = d.xi2 * f2 + d.xi3 * f3;
    const double sll = d.xl2 * f2 + d.xl3 * f3 + d.xl4 * sinzf;
    const double sghl = d.xgh2 * f2 + d.xgh3 * f3 + d.xgh4 * sinzf, shl = d.xh2 * f2 + d.xh3 * f3;
    const double pe = ses + sel, pinc = sis + sil, pl = sls + sll;
    double pgh = sghs + sghl, ph = shs + shl;
    *xinc += pinc;
    *em += pe;
    if (d.xqncl >= 0.2) {
        ph /= sinio;
        pgh -= cosio * ph;
        *omgadf += pgh;
        *xnode += ph;
        *xll += pl;
    } else {
        // Lyddane's modification for low inclinations
        const double sinok = sin(*xnode), cosok = cos(*xnode);
        double alfdp = sinis * sinok, betdp = sinis * cosok;
        const double dalf = ph * cosok + pinc * cosis * sinok, dbet = -ph * sinok + pinc * cosis * cosok;
        alfdp += dalf;
        betdp += dbet;
        *xnode = fmod2p(*xnode);
        double xls = *xll + *omgadf + cosis * *xnode;
        const double dls = pl + pgh - pinc * *xnode * sinis;
        xls += dls;
        const double xnoh = *xnode;
        *xnode = actan(alfdp, betdp);
        if (fabs(xnoh - *xnode) > kPi) *xnode += *xnode < xnoh ? kTwoPi : -kTwoPi;
        *xll += pl;
        *omgadf = xls - *xll - cos(*xinc) * *xnode;
    }
}

void Sgp4::propagate_deep(double ts, double pos[3], double vel[3]) const
{
    const Tle &t = tle_;
    const double ke = xke();
    const double xmdf = t.xmo + xmdot * ts;
    double omgadf = t.omegao + omgdot * ts;
    const double xnoddf = t.xnodeo + xnodot * ts;
    const double tsq = ts * ts;
    double xnode = xnoddf + xnodcf * tsq;
    const double tempa = 1.0 - c1 * ts, tempe = t.bstar * c4 * ts, templ = t2cof * tsq;
    double xn = xnodp, xll = xmdf, em = 0, xinc = 0;
    deep_secular(ts, &xll, &omgadf, &xnode, &em, &xinc, &xn);
    const double a = pow(ke / xn, kTothrd) * tempa * tempa;
    em -= tempe;
    double xmam = xll + xnodp * templ;
    deep_periodic(ts, &em, &xinc, &omgadf, &xnode, &xmam);
    const double e = em;
    const double xl = xmam + omgadf + xnode;
    const double beta = sqrt(1.0 - e * e);
    xn = ke / pow(a, 1.5);
    // long-period periodics
    const double axn = e * cos(omgadf);
    double temp = 1.0 / (a * beta * beta);
    const double xlt = xl + temp * xlcof * axn, ayn = e * sin(omgadf) + temp * aycof;
    // Kepler's equation
    const double capu = fmod2p(xlt - xnode);
    double temp2 = capu, sinepw = 0, cosepw = 0, temp3 = 0, temp4 = 0, temp5 = 0, temp6 = 0;
    for (int i = 0; i < 10; ++i) {
        sinepw = sin(temp2);
        cosepw = cos(temp2);
        temp3 = axn * sinepw;
        temp4 = ayn * cosepw;
        temp5 = axn * cosepw;
        temp6 = ayn * sinepw;
        const double epw = (capu - temp4 + temp3 - temp2) / (1.0 - temp5 - temp6) + temp2;
        if (fabs(epw - temp2) <= 1e-6) break;
        temp2 = epw;
    }
    // short-period preliminary quantities
    const double ecose = temp5 + temp6, esine = temp3 - temp4, elsq = axn * axn + ayn * ayn;
    temp = 1.0 - elsq;
    const double pl = a * temp, r = a * (1.0 - ecose);
    double temp1 = 1.0 / r;
    const double rdot = ke * sqrt(a) * esine * temp1, rfdot = ke * sqrt(pl) * temp1;
    temp2 = a * temp1;
    const double betal = sqrt(temp);
    temp3 = 1.0 / (1.0 + betal);
    const double cosu = temp2 * (cosepw - axn + ayn * esine * temp3);
    const double sinu = temp2 * (sinepw - ayn - axn * esine * temp3);
    const double u = atan2(sinu, cosu);
    const double sin2u = 2.0 * sinu * cosu, cos2u = 2.0 * cosu * cosu - 1.0;
    temp = 1.0 / pl;
    temp1 = kCk2 * temp;
    temp2 = temp1 * temp;
    // short periodics (with the epoch's inclination functions, as in the report)
    const double rk = r * (1.0 - 1.5 * temp2 * betal * x3thm1) + 0.5 * temp1 * x1mth2 * cos2u;
    const double uk = u - 0.25 * temp2 * x7thm1 * sin2u;
    const double xnodek = xnode + 1.5 * temp2 * cosio * sin2u;
    const double xinck = xinc + 1.5 * temp2 * cosio * sinio * cos2u;
    const double rdotk = rdot - xn * temp1 * x1mth2 * sin2u;
    const double rfdotk = rfdot + xn * temp1 * (x1mth2 * cos2u + 1.5 * x3thm1);
    const double sinuk = sin(uk), cosuk = cos(uk), sinik = sin(xinck), cosik = cos(xinck);
    const double sinnok = sin(xnodek), cosnok = cos(xnodek);
    const double xmx = -sinnok * cosik, xmy = cosnok * cosik;
    const double ux = xmx * sinuk + cosnok * cosuk, uy = xmy * sinuk + sinnok * cosuk, uz = sinik * sinuk;
    const double vx = xmx * cosuk - cosnok * sinuk, vy = xmy * cosuk - sinnok * sinuk, vz = sinik * cosuk;
    pos[0] = rk * ux * kXkmper;
    pos[1] = rk * uy * kXkmper;
    pos[2] = rk * uz * kXkmper;
    const double vs = kXkmper / 60.0;
    vel[0] = (rdotk * ux + rfdotk * vx) * vs;
    vel[1] = (rdotk * uy + rfdotk * vy) * vs;
    vel[2] = (rdotk * uz + rfdotk * vz) * vs;
}

LookAngles Sgp4::observe(const Observer &obs, double unix_time_s) const
{
    const double jd = unix_to_jd(unix_time_s);
    double sp[3], sv[3];
    propagate((jd - tle_.epoch_jd) * 1440.0, sp, sv);
    // observer in ECI
    const double lat = obs.lat_deg * kDeg, lon = obs.lon_deg * kDeg, alt = obs.alt_m / 1000.0;
    const double theta = fmod2p(theta_g(jd) + lon);
    const double c = 1.0 / sqrt(1.0 + kF * (kF - 2.0) * sin(lat) * sin(lat));
    const double sq = (1.0 - kF) * (1.0 - kF) * c;
    const double achcp = (kXkmper * c + alt) * cos(lat);
    const double op[3] = {achcp * cos(theta), achcp * sin(theta), (kXkmper * sq + alt) * sin(lat)};
    const double mfactor = kTwoPi * kOmegaE / kSecDay;
    const double ov[3] = {-mfactor * op[1], mfactor * op[0], 0.0};
    const double rx = sp[0] - op[0], ry = sp[1] - op[1], rz = sp[2] - op[2];
    const double vx = sv[0] - ov[0], vy = sv[1] - ov[1], vz = sv[2] - ov[2];
    LookAngles la;
    la.range_km = sqrt(rx * rx + ry * ry + rz * rz);
    la.range_rate_km_s = (rx * vx + ry * vy + rz * vz) / la.range_km;
    // topocentric horizon (south, east, up)
    const double sl = sin(lat), cl = cos(lat), st = sin(theta), ct = cos(theta);
    const double top_s = sl * ct * rx + sl * st * ry - cl * rz;
    const double top_e = -st * rx + ct * ry;
    const double top_z = cl * ct * rx + cl * st * ry + sl * rz;
    double az = atan2(top_e, -top_s);
    if (az < 0) az += kTwoPi;
    la.az_deg = az / kDeg;
    la.el_deg = asin(top_z / la.range_km) / kDeg;
    return la;
}

}  // namespace dpx
