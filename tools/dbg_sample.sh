#!/bin/bash
cd $GRAFT_REPO_ROOT
python3 - <<'PY'
import sys, os, subprocess
sys.path.insert(0,'tests'); sys.path.insert(0,'tests/golden')
import harness
from cases import CASES
CLI='pbsim3_amd/bin/pbsim'
os.makedirs('/tmp/dbg',exist_ok=True)
for case in ['wgs_sample_plain']:
    args=harness.resolve(CASES[case]['args'])
    for ranks in (1,2):
        cmd=[CLI]+args+['--prefix','/tmp/dbg/out%d'%ranks,'--no-gzip']+(['--devices',','.join(['0']*ranks)] if ranks>1 else [])
        try:
            p=subprocess.run(cmd,capture_output=True,text=True,timeout=60,env=dict(os.environ,PBSIM_TRACE='1'))
            print(case,ranks,'rc',p.returncode); print(p.stderr[-3000:])
        except subprocess.TimeoutExpired as e:
            print(case,ranks,'TIMEOUT'); print((e.stderr or b'')[-4000:].decode(errors='replace') if isinstance(e.stderr,bytes) else (e.stderr or '')[-4000:])
PY
