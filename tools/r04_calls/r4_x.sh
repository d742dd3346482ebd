#!/bin/bash
# where does the default line's drop-in figure (2 470) come from when the sweep cell of the same box says 2 810?
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT
show() { python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', d['value'], 'dropin', d['dropin_views_per_s'], 'tl', (d.get('training_like') or {}).get('dropin_views_per_s'), 'fo', (d.get('forward_only') or {}).get('dropin_views_per_s'))"; }
timeout 70 python bench.py --no-cpu-baseline --train-seconds 0 2>/dev/null | tail -1 | show "no-train-legs seg=auto:"
timeout 100 python bench.py --no-cpu-baseline 2>/dev/null | tail -1 | show "full seg=auto:"
GSR_SEG_LEN=256 timeout 100 python bench.py --no-cpu-baseline 2>/dev/null | tail -1 | show "full seg=256:"
