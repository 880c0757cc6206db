import sys, os, torch, torch.distributed as dist, socket
root = os.environ.get('GRAFT_REPO_ROOT', '/root/repo')
sys.path[:0] = [root, os.path.join(root, 'msmc-tts_amd'), os.path.join(root, 'tests')]
torch.cuda.set_device(0)
s = socket.socket(); s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]; s.close()
dist.init_process_group('nccl', init_method='tcp://127.0.0.1:%d' % port, world_size=1, rank=0)
import _parity
from msmctts_amd.distributed import distributed
distributed.apply_gradient_allreduce.__defaults__ = (64 * 1024, None)
from msmctts_amd.hip import convnet
_orig_finish = convnet.ConvBank._finish_backward
def _slow_finish(self, early=False):
    if early and self.w1.is_cuda:
        torch.cuda._sleep(int(2e8))
    return _orig_finish(self, early)
convnet.ConvBank._finish_backward = _slow_finish
# the reducer WITHOUT the stream ordering (round 4's form): the delayed deliveries must now break the golden step
def _launch_unordered(self, b):
    ps = [p for p in b.params if any(p is r for r in b.ready)]
    if ps:
        flat = torch.cat([p.grad.reshape(-1) for p in ps]).to(self.exchange_dtype)
        work = dist.all_reduce(flat, group=self.group, async_op=True)
        self._inflight.append((work, flat, ps))
    b.ready = []; b.streams = []
    b.pending = set(id(p) for p in b.params)
distributed.GradReducer._launch = _launch_unordered
try:
    _parity.check_train_steps('cuda:0', arm_reducer=True)
    print('UNORDERED-REDUCER: golden step still matches (the test does not see the race)')
except AssertionError as e:
    print('UNORDERED-REDUCER: golden step BROKEN as expected:', str(e)[:200])
dist.barrier(); torch.cuda.synchronize(); dist.destroy_process_group()
