#!/bin/bash
# Diagnostic (GPU box): the episode legs of bench.py's extra block on their own (usage: legs_probe.sh [PDB_LIB variant])
[ -n "${1:-}" ] && export PDB_LIB=$1
run() { timeout 200 python3 bench.py --no-cpu-baseline --no-extra "${@:2}" 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('  $1: %.2f M env-steps/s, contact-pass cars %s' % (d['value']/1e6, d.get('contact_pass_cars')))" || echo "  $1: FAILED"; }
run episodes_4096 --workload touge --walls --cars 4096 --episodes --steps 600 --warmup 100 --settle 200
run episodes_4096_reset_free --workload touge --walls --cars 4096 --policy feedback --steps 600 --warmup 100 --settle 200
run episodes_4096_no_body_contacts --workload touge --walls --cars 4096 --policy feedback --steps 600 --warmup 100 --settle 200 --no-body-contacts
run episodes_16384 --workload touge --walls --cars 16384 --episodes --steps 300 --warmup 50 --settle 200
run episodes_16384_reset_free --workload touge --walls --cars 16384 --policy feedback --steps 300 --warmup 50 --settle 200
run headline --steps 3000 --warmup 300
run headline_no_body_contacts --steps 3000 --warmup 300 --no-body-contacts
