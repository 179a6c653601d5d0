#!/bin/bash
# `doppler const` through two pipes (the reference's only I/O mode: main.rs:57-58), DOPPLER_STATS rate, with producers and
# consumers of different speeds:   tools/cli_pipes.sh [BYTES]   -> gpurun_out/cli_pipes.log
N=${1:-4294967296}
OUT=gpurun_out/cli_pipes.log
mkdir -p gpurun_out
EXE=doppler_amd/bin/doppler
BIN=${DPX_TOOLS_BIN:-/tmp/dpx_tools}      # tools/build_tools.sh puts the helper binaries there (outside the tree that ships)
[ -x $BIN/pipe_source ] || tools/build_tools.sh || exit 1
ARGS="const -s 1024000 -i i16 --shift 5000"
: > $OUT
run() {   # label, producer command, consumer command, extra env
  echo "== $1" >> $OUT
  ( eval "$2" | env DOPPLER_STATS=1 $4 $EXE $ARGS 2>> $OUT | eval "$3" ) 2>> $OUT
}
head -c $N /dev/urandom > /dev/shm/dpx_pipes_in.iq 2>/dev/null || dd if=/dev/urandom of=/dev/shm/dpx_pipes_in.iq bs=1M count=$((N >> 20)) 2>/dev/null
echo "plain pipe ceiling (no doppler): pipe_source | pipe_sink" >> $OUT
T0=$(date +%s%N); $BIN/pipe_source $N | $BIN/pipe_sink 2>> $OUT; echo "write -> read: $(( ($(date +%s%N) - T0) / 1000000 )) ms" >> $OUT
T0=$(date +%s%N); $BIN/pipe_source $N vmsplice | $BIN/pipe_sink 2>> $OUT; echo "vmsplice -> read: $(( ($(date +%s%N) - T0) / 1000000 )) ms" >> $OUT
T0=$(date +%s%N); $BIN/pipe_source $N | $BIN/pipe_sink splice 2>> $OUT; echo "write -> splice: $(( ($(date +%s%N) - T0) / 1000000 )) ms" >> $OUT
run "cat file | doppler | cat > /dev/null (64 KiB pipes kept: DOPPLER_NO_PIPE_GROW)" "cat /dev/shm/dpx_pipes_in.iq" "cat > /dev/null" "DOPPLER_NO_PIPE_GROW=1"
run "cat file | doppler | cat > /dev/null" "cat /dev/shm/dpx_pipes_in.iq" "cat > /dev/null" ""
run "dd bs=8M | doppler | dd bs=8M of=/dev/null" "dd if=/dev/shm/dpx_pipes_in.iq bs=8M 2>/dev/null" "dd of=/dev/null bs=8M 2>/dev/null" ""
run "pipe_source (write) | doppler | pipe_sink (read)" "$BIN/pipe_source $N" "$BIN/pipe_sink" ""
run "pipe_source (vmsplice) | doppler (DOPPLER_VMSPLICE=1) | pipe_sink (splice)" "$BIN/pipe_source $N vmsplice" "$BIN/pipe_sink splice" "DOPPLER_VMSPLICE=1"
echo "== input side alone: pipe_source (vmsplice) | doppler > /dev/null" >> $OUT
( $BIN/pipe_source $N vmsplice | env DOPPLER_STATS=1 $EXE $ARGS > /dev/null ) 2>> $OUT
echo "== output side alone: doppler < file | pipe_sink (splice)" >> $OUT
( env DOPPLER_STATS=1 $EXE $ARGS < /dev/shm/dpx_pipes_in.iq | $BIN/pipe_sink splice ) 2>> $OUT
echo "== output side alone, write(): DOPPLER_NO_VMSPLICE=1 doppler < file | pipe_sink (splice)" >> $OUT
( env DOPPLER_STATS=1 DOPPLER_NO_VMSPLICE=1 $EXE $ARGS < /dev/shm/dpx_pipes_in.iq | $BIN/pipe_sink splice ) 2>> $OUT
run "output by write() (DOPPLER_NO_VMSPLICE): pipe_source (vmsplice) | doppler | pipe_sink (splice)" "$BIN/pipe_source $N vmsplice" "$BIN/pipe_sink splice" "DOPPLER_NO_VMSPLICE=1"
for sl in 1048576 16777216; do
  run "pipe_source (vmsplice) | doppler | pipe_sink (splice), slab $sl" "$BIN/pipe_source $N vmsplice" "$BIN/pipe_sink splice" "DOPPLER_SLAB_BYTES=$sl"
done
rm -f /dev/shm/dpx_pipes_in.iq
grep -E "^==|Msamples|pipe buffers|pipe_sink| ms$" $OUT
