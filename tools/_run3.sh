python tools/host_call_cost.py
