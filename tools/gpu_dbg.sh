#!/bin/bash
for d in ${DBGS:-0 8 16 24}; do echo "QRL_DBG=$d"; QRL_DBG=$d timeout 300 python bench.py --no-extra --steps 5 --warmup 1 2>&1 | grep -o '"kernel_ms": [0-9.]*'; done
