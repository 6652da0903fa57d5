python -m pytest tests -m gpu -x -q --durations=8 2>&1 | grep -v "^$" | tail -20
