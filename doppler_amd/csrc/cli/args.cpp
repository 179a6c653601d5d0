// args.cpp — see args.h.
#include "args.h"

#include <errno.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <map>
#include <vector>

#include "../host/orbit.h"

namespace dpx {

const char *datatype_name(DataType t) { return t == DataType::F32 ? "f32" : "i16"; }   // usage.rs:44-51

// usage.rs:85-115.  Same acceptance rule: all three keys must appear, each "key=value" item
// separated by commas, values parsed as f64.
bool parse_location(const std::string &s, Location *out, std::string *err)
{
    if (s.find("lat") == std::string::npos || s.find("lon") == std::string::npos || s.find("alt") == std::string::npos) {
        *err = "--location should be defined as: lat=58.64560,lon=23.15163,alt=8";
        return false;
    }
    bool hl = false, ho = false, ha = false;
    size_t pos = 0;
    while (pos <= s.size()) {
        size_t c = s.find(',', pos);
        if (c == std::string::npos) c = s.size();
        const std::string item = s.substr(pos, c - pos);
        const size_t eq = item.find('=');
        if (eq != std::string::npos) {
            std::string val = item.substr(eq + 1);
            const size_t eq2 = val.find('=');               // `s.split("=").nth(1)`
            if (eq2 != std::string::npos) val = val.substr(0, eq2);
            char *end = nullptr;
            errno = 0;
            const double v = strtod(val.c_str(), &end);
            const bool ok = !val.empty() && end && *end == '\0' && errno == 0;
            if (item.find("lat") != std::string::npos) { if (ok) { out->lat = v; hl = true; } }
            else if (item.find("lon") != std::string::npos) { if (ok) { out->lon = v; ho = true; } }
            else if (item.find("alt") != std::string::npos) { if (ok) { out->alt = v; ha = true; } }
        }
        pos = c + 1;
    }
    if (hl && ho && ha) return true;
    *err = s + " isn't a valid value for --location\n\t[use as: lat=58.64560,lon=23.15163,alt=8]";
    return false;
}

namespace {

struct Flag {
    const char *name;      // clap argument name (upper case in the reference)
    const char *longf;
    char shortf;           // 0 = none
    bool required;
    bool datatype;         // possible_values ["i16", "f32"]
    const char *help;
};

const Flag kConstFlags[] = {
    {"SAMPLERATE", "samplerate", 's', true, false, "IQ data samplerate"},
    {"INTYPE", "intype", 'i', true, true, "IQ data input type"},
    {"OUTTYPE", "outtype", 'o', false, true, "IQ data output type"},
    {"SHIFT", "shift", 0, true, false, "frequency shift in Hz"},
    // extension of this build (the reference has no such flag): see args.h
    {"GPUS", "gpus", 0, false, false, "[extension] number of GPUs to spread the stream over (default 1, or $DOPPLER_GPUS)"},
    {"GATHER", "gather", 0, false, false, "[extension] with --gpus N: how outputs reach the host: d2h = every GPU over its own PCIe link (default), rccl = over xGMI into the first GPU, from there"},
};
const Flag kTrackFlags[] = {
    {"SAMPLERATE", "samplerate", 's', true, false, "IQ data samplerate"},
    {"INTYPE", "intype", 'i', true, true, "IQ data type"},
    {"OUTTYPE", "outtype", 'o', false, true, "IQ data output type"},
    {"TLEFILE", "tlefile", 0, true, false, "TLE file: eg. http://www.celestrak.com/NORAD/elements/cubesat.txt"},
    {"TLENAME", "tlename", 0, true, false, "TLE name in TLE file: eg. ESTCUBE 1"},
    {"LOCATION", "location", 0, true, false, "Observer location (lat=<deg>,lon=<deg>,alt=<m>): eg. lat=58.64560,lon=23.15163,alt=8"},
    {"TIME", "time", 0, false, false, "Observation start time in UTC Y-m-dTH:M:S: eg. 2015-05-13T14:28:48. If not specified current time is used"},
    {"FREQUENCY", "frequency", 0, true, false, "Satellite transmitter frequency in Hz"},
    {"OFFSET", "offset", 0, false, false, "Constant frequency shift in Hz. Can be used to compensate constant offset"},
    // extension of this build (the reference has no such flag): see args.h
    {"RANGERATEFILE", "range-rate-file", 0, false, false, "[extension] text file with one range rate (km/s) per whole second; replaces --tlefile/--tlename/--location"},
    {"GPUS", "gpus", 0, false, false, "[extension] number of GPUs to spread the stream over (default 1, or $DOPPLER_GPUS)"},
    {"GATHER", "gather", 0, false, false, "[extension] with --gpus N: how outputs reach the host: d2h = every GPU over its own PCIe link (default), rccl = over xGMI into the first GPU, from there"},
};

void usage(FILE *f, const char *sub, const Flag *flags, size_t n)
{
    fprintf(f, "doppler-%s\n%s\n\nUSAGE:\n    doppler %s [OPTIONS]\n\nOPTIONS:\n", sub,
            strcmp(sub, "const") == 0 ? "Constant shift mode" : "Doppler tracking mode", sub);
    for (size_t i = 0; i < n; ++i) {
        char sh[8] = "   ";
        if (flags[i].shortf) snprintf(sh, sizeof(sh), "-%c,", flags[i].shortf);
        fprintf(f, "    %s --%s <%s>    %s%s%s\n", sh, flags[i].longf, flags[i].name, flags[i].help,
                flags[i].datatype ? " [values: i16, f32]" : "", flags[i].required ? "" : " (optional)");
    }
}

void top_usage(FILE *f)
{
    fprintf(f, "doppler (MI355X build)\nCompensates IQ data stream doppler shift based on TLE information, also can be used "
               "for doing constant baseband shifting\n\nUSAGE:\n    doppler [SUBCOMMAND]\n\nFLAGS:\n    -h, --help       Prints help "
               "information\n    -V, --version    Prints version information\n\nSUBCOMMANDS:\n    const    Constant shift mode\n"
               "    track    Doppler tracking mode\n");
}

int clap_error(const char *fmt, const std::string &a = "", const std::string &b = "")
{
    fprintf(stderr, "error: ");
    fprintf(stderr, fmt, a.c_str(), b.c_str());
    fprintf(stderr, "\n\nFor more information try --help\n");
    return 1;
}

template <typename T> bool parse_int(const std::string &s, T *out)
{
    if (s.empty()) return false;
    errno = 0;
    char *end = nullptr;
    if (sizeof(T) == 4 && T(-1) < T(0)) {
        const long long v = strtoll(s.c_str(), &end, 10);
        if (errno || *end || v < INT32_MIN || v > INT32_MAX) return false;
        *out = (T)v;
    } else {
        if (s[0] == '-') return false;
        const unsigned long long v = strtoull(s.c_str(), &end, 10);
        if (errno || *end || v > UINT32_MAX) return false;
        *out = (T)v;
    }
    return true;
}

}  // namespace

int parse_args(int argc, char **argv, CommandArgs *out, bool *exit_now)
{
    *exit_now = true;
    if (argc < 2) {
        fprintf(stderr, "no arguments provided, try with doppler -h\n");     // usage.rs:331-333
        return 1;
    }
    const std::string sub = argv[1];
    if (sub == "-h" || sub == "--help" || sub == "help") { top_usage(stdout); return 0; }
    if (sub == "-V" || sub == "--version") { printf("doppler 1.1.10-mi355x\n"); return 0; }
    const Flag *flags;
    size_t nflags;
    if (sub == "const") { flags = kConstFlags; nflags = sizeof(kConstFlags) / sizeof(Flag); out->mode = Mode::Const; }
    else if (sub == "track") { flags = kTrackFlags; nflags = sizeof(kTrackFlags) / sizeof(Flag); out->mode = Mode::Track; }
    else return clap_error("Found argument '%s' which wasn't expected, or isn't valid in this context", sub);

    std::map<std::string, std::string> val;
    for (int i = 2; i < argc; ++i) {
        std::string a = argv[i];
        if (a == "-h" || a == "--help") { usage(stdout, sub.c_str(), flags, nflags); return 0; }
        const Flag *fl = nullptr;
        std::string inline_val;
        bool has_inline = false;
        if (a.rfind("--", 0) == 0) {
            std::string name = a.substr(2);
            const size_t eq = name.find('=');
            if (eq != std::string::npos) { inline_val = name.substr(eq + 1); name = name.substr(0, eq); has_inline = true; }
            for (size_t k = 0; k < nflags; ++k) if (name == flags[k].longf) fl = &flags[k];
        } else if (a.size() >= 2 && a[0] == '-') {
            for (size_t k = 0; k < nflags; ++k) if (flags[k].shortf && a[1] == flags[k].shortf) fl = &flags[k];
            if (fl && a.size() > 2) { inline_val = a.substr(a[2] == '=' ? 3 : 2); has_inline = true; }
        }
        if (!fl) return clap_error("Found argument '%s' which wasn't expected, or isn't valid in this context", a);
        std::string v;
        if (has_inline) v = inline_val;
        else if (i + 1 < argc) v = argv[++i];      // AllowLeadingHyphen: the next token is the value whatever it looks like
        else return clap_error("The argument '--%s <%s>' requires a value but none was supplied", fl->longf, fl->name);
        if (val.count(fl->name)) return clap_error("The argument '--%s <%s>' was provided more than once, but cannot be used multiple times", fl->longf, fl->name);
        if (fl->datatype && v != "i16" && v != "f32")
            return clap_error("'%s' isn't a valid value for '--%s'\n\t[values: i16, f32]", v, fl->longf);
        val[fl->name] = v;
    }
    const bool table_mode = val.count("RANGERATEFILE") != 0;
    std::vector<std::string> missing;
    for (size_t k = 0; k < nflags; ++k) {
        const bool orbit_flag = !strcmp(flags[k].name, "TLEFILE") || !strcmp(flags[k].name, "TLENAME") || !strcmp(flags[k].name, "LOCATION");
        if (flags[k].required && !val.count(flags[k].name) && !(table_mode && orbit_flag))
            missing.push_back(std::string("--") + flags[k].longf + " <" + flags[k].name + ">");
    }
    if (!missing.empty()) {
        std::string m;
        for (const std::string &x : missing) m += "\n    " + x;
        return clap_error("The following required arguments were not provided:%s", m);
    }
    auto bad = [&](const char *name) {
        return clap_error("Invalid value: The argument '%s' isn't a valid value", val[name]);   // value_t_or_exit!
    };
    if (!parse_int<uint32_t>(val["SAMPLERATE"], &out->samplerate)) return bad("SAMPLERATE");
    if (val.count("GPUS") && (!parse_int<uint32_t>(val["GPUS"], &out->gpus) || out->gpus < 1 || out->gpus > 64)) return bad("GPUS");
    if (val.count("GATHER")) {
        if (val["GATHER"] == "rccl") out->gather_rccl = true;
        else if (val["GATHER"] != "d2h") return bad("GATHER");
    }
    out->inputtype = val["INTYPE"] == "f32" ? DataType::F32 : DataType::I16;
    out->outputtype = val.count("OUTTYPE") ? (val["OUTTYPE"] == "f32" ? DataType::F32 : DataType::I16) : out->inputtype;
    if (out->mode == Mode::Const) {
        if (!parse_int<int32_t>(val["SHIFT"], &out->shift)) return bad("SHIFT");
    } else {
        if (val.count("OFFSET")) {
            if (!parse_int<int32_t>(val["OFFSET"], &out->offset)) return bad("OFFSET");
            out->has_offset = true;
        }
        if (val.count("TIME")) {
            if (!parse_utc(val["TIME"].c_str(), &out->time_unix)) {           // usage.rs:303-312
                fprintf(stderr, "--time should be defined in Y-m-dTH:M:S format: eg. 2015-05-13T14:28:48\n");
                return 1;
            }
            out->has_time = true;
        }
        if (!parse_int<uint32_t>(val["FREQUENCY"], &out->frequency)) return bad("FREQUENCY");
        if (table_mode) {
            out->range_rate_file = val["RANGERATEFILE"];
        } else {
            out->tlefile = val["TLEFILE"];
            out->tlename = val["TLENAME"];
            std::string err;
            if (!parse_location(val["LOCATION"], &out->location, &err)) {       // usage.rs:320-327
                fprintf(stderr, "%s.\n", err.c_str());
                return 1;
            }
        }
    }
    *exit_now = false;
    return 0;
}

}  // namespace dpx
