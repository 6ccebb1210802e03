"""PCIe-inclusive cost of handing a HOST trace to the device: nx_upload_columns over 2^log-row columns."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import nexus_zkvm_amd as nz
log = int(sys.argv[1]) if len(sys.argv) > 1 else 22
ncols = int(sys.argv[2]) if len(sys.argv) > 2 else 64
be = nz.HipBackend(0)
host = [np.random.default_rng(i).integers(0, nz.P, 1 << log, dtype=np.uint32) for i in range(ncols)]
res = {}
for name, co in (("coset_order_permuted", True), ("as_is", False)):
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter(); d = be.upload_columns(host, coset_order=co); be.sync(); best = min(best, time.perf_counter() - t0); d.free()
    res[name + "_GBs"] = ncols * (4 << log) / best / 1e9
    res[name + "_ms"] = best * 1e3
t0 = time.perf_counter(); d = be.columns_from_host(np.stack(host)); be.sync(); res["pageable_hipMemcpy_GBs"] = ncols * (4 << log) / (time.perf_counter() - t0) / 1e9
print(json.dumps({"log_size": log, "columns": ncols, **res}))
