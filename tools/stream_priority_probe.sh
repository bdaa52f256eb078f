python - <<'PY'
import torch
try:
    print("priority_range", torch.cuda.Stream.priority_range())
except Exception as e:
    print("no priority_range:", e)
for p in (-2,-1,0,1,2):
    try:
        s=torch.cuda.Stream(priority=p); print("priority", p, "ok ->", s.priority)
    except Exception as e:
        print("priority", p, "fails:", str(e)[:80])
PY
