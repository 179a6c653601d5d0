for shape in 4,2 2,2; do echo "== shape $shape"; DPX_WALK_SHAPE=$shape python tools/track_probe.py 2>&1 | grep auto | grep -v "4096\|16384\|same"; done
