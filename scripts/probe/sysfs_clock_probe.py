"""Which sysfs / hwmon files report the shader clock UNDER LOAD on this box?  Reads every candidate while a kernel stream keeps the GPU busy."""
import glob
import threading
import time

import torch

paths = sorted(glob.glob("/sys/class/drm/card*/device/pp_dpm_*") + glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*/freq*_input")
               + glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*/freq*_label") + glob.glob("/sys/class/drm/card*/device/gpu_busy_percent")
               + glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*/power*_average") + glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*/power*_input"))


def dump(tag):
    print("==", tag)
    for p in paths:
        try:
            print(p, "->", open(p).read().strip().replace("\n", " | "))
        except OSError as e:
            print(p, "ERR", e)


dump("idle")
x = torch.randn(8192, 8192, device="cuda")
stop = False


def busy():
    while not stop:
        for _ in range(20):
            torch.mm(x, x)
        torch.cuda.synchronize()


t = threading.Thread(target=busy)
t.start()
time.sleep(1.0)
dump("under load (fp32 GEMM stream)")
time.sleep(0.5)
dump("under load again")
stop = True
t.join()
