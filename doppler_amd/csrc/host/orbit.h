// orbit.h — host-side range-rate provider for `doppler track`.
//
// The reference takes range rate from rust-gpredict -> C libgpredict (reference
// src/main.rs:149,162-163: Predict::new / predict.update / predict.sat.range_rate_km_sec).
// Neither the crate nor the library is in /root/reference, so nothing here can be
// compared with it: ORBIT PARITY UNPINNED.  What is implemented is the published
// algorithm those libraries are built on — NORAD SGP4 (Spacetrack Report #3,
// near-earth model, WGS-72 constants) and the usual geodetic-observer range-rate
// computation.  Deep-space element sets (period >= 225 min) are rejected.
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
    // false if the element set needs the deep-space model
    bool init(const Tle &tle, std::string *err);
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
};

double unix_to_jd(double unix_time_s);
// "%Y-%m-%dT%H:%M:%S" (reference src/usage.rs:303) interpreted as UTC; false if malformed
bool parse_utc(const char *s, int64_t *unix_time_s);

}  // namespace dpx
