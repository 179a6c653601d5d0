// args.h — command-line surface of `doppler`.
//
// Mirrors reference src/usage.rs:117-337 (clap 2 builder): subcommands `const` and `track`,
// the same long/short flags, required/optional status, possible values and the leading-hyphen
// rule that lets `--shift -5000` parse (usage.rs:127,161).  Errors print a message to stderr
// and exit with status 1, as clap's `value_t_or_exit!` / the explicit `exit(1)` calls do.
#pragma once
#include <stdint.h>

#include <string>

namespace dpx {

enum class Mode { Const, Track };          // usage.rs:33-36
enum class DataType { F32, I16 };          // usage.rs:38-42

struct Location {                          // usage.rs:53-59
    double lat = 0, lon = 0, alt = 0;
};

struct CommandArgs {                       // usage.rs:61-83
    Mode mode = Mode::Const;
    uint32_t samplerate = 0;
    DataType inputtype = DataType::I16;
    DataType outputtype = DataType::I16;   // defaults to inputtype (usage.rs:268-270)
    // const
    int32_t shift = 0;
    // track
    std::string tlefile, tlename;
    Location location;
    bool has_time = false;
    int64_t time_unix = 0;                 // --time %Y-%m-%dT%H:%M:%S, UTC (usage.rs:303)
    uint32_t frequency = 0;
    bool has_offset = false;
    int32_t offset = 0;
    // extension (not in the reference): whole-second range-rate table instead of TLE propagation
    std::string range_rate_file;
    // extension (not in the reference): GPUs to spread the stream over; 0 = not given (1, or $DOPPLER_GPUS)
    uint32_t gpus = 0;
    // extension (not in the reference): --gather rccl: outputs of GPUs 1..N-1 over RCCL into the first GPU (the form
    // BASELINE.json's north_star names); default: every GPU copies its own output to the host
    bool gather_rccl = false;
};

// usage.rs:85-115 parse_location: "lat=58.64560,lon=23.15163,alt=8"
bool parse_location(const std::string &s, Location *out, std::string *err);

// Returns 0 and fills `out`, or prints the error / help text and returns the exit status to use
// (1 for errors as in the reference, 0 after -h/--help/-V).
int parse_args(int argc, char **argv, CommandArgs *out, bool *exit_now);

const char *datatype_name(DataType t);

}  // namespace dpx
