set -e
timeout -k 10 600 python -m pytest tests/test_hip_scene.py -x -q -m gpu -k "trunk or oracle_on_seeded or buffers" 2>&1 | tail -4
bash tools/dbg/tr_variants.sh "-DTR_DBG=0"
