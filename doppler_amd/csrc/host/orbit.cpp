// orbit.cpp — see orbit.h.  NORAD SGP4 (Spacetrack Report #3, near-earth), WGS-72.
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
    if (kTwoPi / xnodp >= 225.0) {
        if (err) *err = "deep-space element set (period >= 225 min): SDP4 is not implemented";
        return false;
    }
    simple_ = (aodp * (1.0 - t.eo) / kAe) < (220.0 / kXkmper + kAe);
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
