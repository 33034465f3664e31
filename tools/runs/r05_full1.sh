#!/bin/bash
# round 5: the whole GPU suite on the library as it stands (PV 4 pivot blocks, backward lists from the root side, two-level PCG preconditioner, ...)
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/r05_full1
mkdir -p $OUT
cd $ROOT
timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -15 > $OUT/pytest.txt; cat $OUT/pytest.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
