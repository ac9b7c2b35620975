import ctypes, torch
hip = ctypes.CDLL('libamdhip64.so')
torch.cuda.init()
for name, idx in (('MaxSharedMemoryPerBlock', None),):
    pass
p = torch.cuda.get_device_properties(0)
print('torch props:', getattr(p, 'shared_memory_per_block', None), getattr(p, 'shared_memory_per_block_optin', None), getattr(p, 'shared_memory_per_multiprocessor', None), p.gcnArchName)
