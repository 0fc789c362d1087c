#!/bin/bash
# end of round 3: smoke, the final measurement set (as r3_u.sh) on the final build
set -e
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
bash tools/gpu/r3_u.sh
