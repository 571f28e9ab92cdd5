import os, sys, subprocess
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
import natural
from PIL import Image
from test_gpu_host_cpp import _write_config_cfg, _env
d = '/tmp/clidbg'; os.makedirs(d, exist_ok=True)
files = []
for k, v in enumerate(natural.config_views(1, 2)):
    p = f'{d}/{k:02d}.png'; Image.fromarray(v).save(p); files.append(p)
_write_config_cfg(f'{d}/config.cfg', LAZY_READ=0, CYLINDER=1, ESTIMATE_CAMERA=0, ORDERED_INPUT=1)
env = _env(); env['OPENPANO_TEST_SEED'] = '38'
for thr in ('1', '8'):
    env['OMP_NUM_THREADS'] = thr
    r = subprocess.run([os.path.abspath('oracle/_ref/image-stitching-hipfast')] + files, capture_output=True, text=True, env=env, cwd=d)
    print('threads', thr, 'rc', r.returncode)
    print(r.stderr[-2500:])
