// orbit.h — host-side range-rate provider for `doppler track`.
//
// The reference takes range rate from rust-gpredict -> C libgpredict (reference
// src/main.rs:149,162-163: Predict::new / predict.update / predict.sat.range_rate_km_sec).
// Neither the crate nor the library is in /root/reference, so nothing here can be
// compared with it: ORBIT PARITY UNPINNED.  What is implemented is the published
// algorithm those libraries are built on — NORAD SGP4 and, for element sets with a
// period of 225 minutes or more (Molniya, GPS, geostationary ...), SDP4 with its
// deep-space subroutine (Spacetrack Report #3, WGS-72 constants): lunar and solar
// secular and periodic terms, and the integrated resonance terms of 12-hour and
// 24-hour orbits — plus the usual geodetic-observer range-rate computation.
// One deliberate difference from the report's code: it recomputes the lunar-solar
// periodics only when the time has moved by 30 minutes since they were last
// computed (a speed-up of 1980 that makes a result depend on the calls before it);
// here they are evaluated at every call.
#pragma once
#include <stdint.h>

#include <string>

namespace dpx {

struct Tle {
    std::string name;
    double epoch_jd = 0;     // Julian date of the element set epoch
    double bstar = 0;        // 1 / earth radii
    double xincl = 0;        // rad
    double xnodeo = 0;       // RAAN, rad
    double eo = 0;           // eccentricity
    double omegao = 0;       // argument of perigee, rad
    double xmo = 0;          // mean anomaly, rad
    double xno = 0;          // mean motion, rad / min
};

// Parse a two-line element set (lines without the name line). Returns false on malformed input.
bool tle_parse(const char *line1, const char *line2, Tle *out, std::string *err);

// Find `name` in a TLE file with 3-line entries (name, line 1, line 2); like
// gpredict's Tle::from_file(name, file) (reference src/main.rs:141).
bool tle_from_file(const char *path, const char *name, Tle *out, std::string *err);

struct Observer {       // reference src/usage.rs:55-59 Location {lat, lon, alt}: degrees, degrees, metres
    double lat_deg = 0, lon_deg = 0, alt_m = 0;
};

struct LookAngles {
    double az_deg = 0, el_deg = 0, range_km = 0, range_rate_km_s = 0;
};

class Sgp4 {
public:
    // chooses the near-earth or the deep-space model from the period; false only for unusable elements
    bool init(const Tle &tle, std::string *err);
    bool deep_space() const { return deep_; }
    // ECI position (km) and velocity (km/s) at `tsince` minutes after the epoch
    void propagate(double tsince_min, double pos[3], double vel[3]) const;
    // what predict.update(time) exposes as predict.sat.* (reference src/main.rs:162-173)
    LookAngles observe(const Observer &obs, double unix_time_s) const;
    double epoch_jd() const { return tle_.epoch_jd; }

private:
    Tle tle_;
    bool simple_ = false;
    double aodp = 0, xnodp = 0, cosio = 0, sinio = 0, x3thm1 = 0, x1mth2 = 0, x7thm1 = 0, eta = 0;
    double c1 = 0, c4 = 0, c5 = 0, d2 = 0, d3 = 0, d4 = 0, xmdot = 0, omgdot = 0, xnodot = 0;
    double omgcof = 0, xmcof = 0, xnodcf = 0, t2cof = 0, t3cof = 0, t4cof = 0, t5cof = 0;
    double xlcof = 0, aycof = 0, delmo = 0, sinmo = 0;
    // ---- deep space (SDP4)
    bool deep_ = false;
    struct Deep {
        double thgr = 0, xnq = 0, xqncl = 0, omegaq = 0, zmol = 0, zmos = 0;
        double sse = 0, ssi = 0, ssl = 0, ssg = 0, ssh = 0;
        double se2 = 0, si2 = 0, sl2 = 0, sgh2 = 0, sh2 = 0, se3 = 0, si3 = 0, sl3 = 0, sgh3 = 0, sh3 = 0, sl4 = 0, sgh4 = 0;
        double ee2 = 0, e3 = 0, xi2 = 0, xi3 = 0, xl2 = 0, xl3 = 0, xl4 = 0, xgh2 = 0, xgh3 = 0, xgh4 = 0, xh2 = 0, xh3 = 0;
        double d2201 = 0, d2211 = 0, d3210 = 0, d3222 = 0, d4410 = 0, d4422 = 0, d5220 = 0, d5232 = 0, d5421 = 0, d5433 = 0;
        double del1 = 0, del2 = 0, del3 = 0, fasx2 = 0, fasx4 = 0, fasx6 = 0, xlamo = 0, xfact = 0;
        bool resonant = false, synchronous = false;
    } dp_;
    void deep_init();
    void deep_secular(double t, double *xll, double *omgadf, double *xnode, double *em, double *xinc, double *xn) const;
    void deep_periodic(double t, double *em, double *xinc, double *omgadf, double *xnode, double *xll) const;
    void propagate_deep(double tsince_min, double pos[3], double vel[3]) const;
};

double unix_to_jd(double unix_time_s);
// "%Y-%m-%dT%H:%M:%S" (reference src/usage.rs:303) interpreted as UTC; false if malformed
bool parse_utc(const char *s, int64_t *unix_time_s);

}  // namespace dpx
