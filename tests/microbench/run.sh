#!/bin/bash
cd "$(dirname "$0")"
for b in bin/mb_*; do timeout -k 5 120 $b ${MB_ARGS:-4 100} || echo "$b failed rc=$?"; done
