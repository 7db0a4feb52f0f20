# determinism at HEAD (GPU box): 200 full-size configs[2] renders next to a busy stream (the classify pass on the step table),
# 40 of the same frame through the one-pass kernel WITH the warp (256 x 256: it is slow), 100,000 replays of the 2-frame step
mkdir -p gpurun_out/r06
echo "== staged renderer, 200 renders of a 1024^2 configs[2] frame"; timeout 900 python tools/exp/render_repeat.py 200 1024 2>&1 | tail -2
echo "== one-pass kernel with the warp, 40 renders of a 256^2 frame"; ANR_ONE_PASS=1 timeout 900 python tools/exp/render_repeat.py 40 256 2>&1 | tail -2
echo "== race_hunt 100000 replays, 2 frames"; timeout 1200 python tools/exp/race_hunt.py 100000 2 2>&1 | tail -1
