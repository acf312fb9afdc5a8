cd $GRAFT_REPO_ROOT
for d in /sys/bus/pci/devices/*; do :; done
python - <<PY
import torch
p=torch.cuda.get_device_properties(0)
bdf="%04x:%02x:%02x.0"%(getattr(p,"pci_domain_id",0),p.pci_bus_id,p.pci_device_id)
print(bdf)
try: print(open("/sys/bus/pci/devices/%s/numa_node"%bdf).read())
except Exception as e: print("no numa_node:",e)
PY
timeout 600 python -m pytest tests/test_devices_native_gpu.py -q -x -m gpu -k "shard_load or lone_worker" 2>&1 | tail -3
