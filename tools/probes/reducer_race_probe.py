"""GPU probe (round 5): tests/_parity.py check_reducer_stream_order -- a stock module and a convolution bank in one reducer
bucket, the bank's early gradient delivery held back 0.1 s on its side stream -- with the reducer of ROUND 4 (no stream ordering
in ``_launch``).  The product's reducer passes the same scenario
(tests/test_gpu_parity.py::test_reducer_waits_for_the_stream_a_bank_delivered_its_gradients_on); this one must not."""
import os
import socket
import sys

import torch
import torch.distributed as dist

root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [root, os.path.join(root, 'msmc-tts_amd'), os.path.join(root, 'tests')]
torch.cuda.set_device(0)
s = socket.socket()
s.bind(('127.0.0.1', 0))
port = s.getsockname()[1]
s.close()
dist.init_process_group('nccl', init_method='tcp://127.0.0.1:%d' % port, world_size=1, rank=0)
import _parity  # noqa: E402
from msmctts_amd.distributed import distributed  # noqa: E402


def _launch_unordered(self, b):                   # round 4: whichever stream completes the bucket concatenates at once
    ps = [p for p in b.params if any(p is r for r in b.ready)]
    if ps:
        flat = torch.cat([p.grad.reshape(-1) for p in ps]).to(self.exchange_dtype)
        work = dist.all_reduce(flat, group=self.group, async_op=True)
        self._inflight.append((work, flat, ps))
    b.ready, b.streams = [], []
    b.pending = set(id(p) for p in b.params)


distributed.GradReducer._launch = _launch_unordered
worst = _parity.check_reducer_stream_order('cuda:0')
print('UNORDERED-REDUCER: largest gradient deviation %.3e (%s)' % (worst, 'the race shows' if not worst < 1e-5 else 'the race does NOT show'))
dist.barrier()
torch.cuda.synchronize()
dist.destroy_process_group()
