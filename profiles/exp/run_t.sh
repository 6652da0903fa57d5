python -m pytest tests -m gpu -x -q -k "crop_pack_all_window" 2>&1 | tail -5
