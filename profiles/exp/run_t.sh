python -m pytest tests -m gpu -x -q -k "ddp_two or handmade" 2>&1 | grep -v "^$" | tail -12
