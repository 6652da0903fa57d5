python -m pytest tests -m gpu -x -q -s -k "headline or zero_detections or batch64 or reference_loop_unchanged or low_precision or deviation" 2>&1 | grep -v "^$" | tail -30
