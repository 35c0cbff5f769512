import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print({k:d[k] for k in ("value","ms_per_step","step_ms","host_enqueue_ms_per_step","loss")}, d["config"]["step_mode"])
