import glob, os, torch
p = torch.cuda.get_device_properties(0)
print("props:", [a for a in dir(p) if "pci" in a.lower()], getattr(p,"pci_bus_id",None), getattr(p,"pci_device_id",None), getattr(p,"pci_domain_id",None))
for c in sorted(glob.glob("/sys/class/drm/card*/device")):
    try:
        real = os.path.realpath(c)
        v = open(c+"/vendor").read().strip()
        print(c, real, v)
        for f in ("pp_dpm_sclk","pp_dpm_mclk","gpu_busy_percent","unique_id"):
            fp=c+"/"+f
            if os.path.exists(fp):
                try: print("  ",f, repr(open(fp).read()[:200]))
                except Exception as e: print("  ",f,"ERR",e)
        for h in glob.glob(c+"/hwmon/hwmon*"):
            for f in sorted(os.listdir(h)):
                if f.startswith(("power","temp","freq")) and (f.endswith("_input") or f.endswith("_average") or f.endswith("_label") or f.endswith("_cap")):
                    try: print("  ",h.split("/")[-1],f, open(h+"/"+f).read().strip())
                    except Exception as e: print("  ",f,"ERR",e)
    except Exception as e:
        print(c,"ERR",e)
print(os.environ.get("HIP_VISIBLE_DEVICES"), os.environ.get("ROCR_VISIBLE_DEVICES"))
import subprocess
try: print(subprocess.run(["rocm-smi","--showclocks","--showpower","--showtemp","--json"],capture_output=True,text=True,timeout=30).stdout[:1500])
except Exception as e: print("rocm-smi ERR", e)
