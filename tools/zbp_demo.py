import os, subprocess, sys, tempfile
import numpy as np
sys.path.insert(0, os.getcwd())
from oracle import zbp as ozbp
from ogl_beamforming_amd import configs
acq = configs.config(2, 1.0)
bp = acq.bp
A = bp.acquisition_count
rf = np.ascontiguousarray(acq.rf)
raw = ozbp.write_v2(ozbp.RCA_TPW, int(bp.data_kind), int(bp.decode_mode), 0,
                    (bp.raw_data_dimensions[0], bp.raw_data_dimensions[1]), bp.sample_count, bp.channel_count, A,
                    tuple(bp.xdc_element_pitch), np.array(bp.xdc_transform[:], np.float32), bp.speed_of_sound,
                    bp.sampling_frequency, bp.demodulation_frequency, bp.time_offset,
                    ("sine", 2.0, bp.demodulation_frequency), tilting_angles=list(bp.steering_angles[:A]),
                    orientations=list(bp.transmit_receive_orientations[:A]), data=rf.tobytes())
d = tempfile.mkdtemp()
p = os.path.join(d, "cfg2.bp")
open(p, "wb").write(raw)
print(len(raw) / 1e6, "MB file")
r = subprocess.run(["./ogl_beamforming_amd/ogl_beamformer_throughput", "--frames", "200", "--lateral", "-0.0128", "0.0128",
                    "--axial", "0.008", "0.05", p], capture_output=True, text=True)
print(r.stdout, r.stderr)
