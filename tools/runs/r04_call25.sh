#!/bin/bash
# round 4, call 25: the summary line of a slice of the suite on the final library (the full run's line was cut off by `tail`)
timeout 125 python -m pytest tests -m gpu -q -x -k "headline_config_camera_parity or ranks_equal_one_rank or (fuzz_slice and std-210) or long_items or packed or replay or adapter" 2>&1 | grep -E "passed|failed|error" | tail -3
