timeout -k 10 200 python tools/srch_probe.py 2>&1 | grep -E "search32|level" | sed 's/.k_prepare.*k_order_search/ search/'
timeout -k 10 200 python tools/srch_probe_l10.py 2>&1 | grep search | sed 's/.k_prepare.*k_order_search/ search/'
