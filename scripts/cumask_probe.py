"""CU-masked streams (hipExtStreamCreateWithCUMask): which XCDs do their workgroups land on, and
what do the RGB trunk on 7 XCDs + the depth trunk on the 8th cost against today's shared 8?

    hipcc --offload-arch=gfx950 -shared -fPIC scripts/xcc_probe.hip -o /tmp/xcc_probe.so
    python scripts/cumask_probe.py
"""
import collections
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

hip = C.CDLL("libamdhip64.so")
probe = C.CDLL("/tmp/xcc_probe.so")
dev = torch.device("cuda:0")
torch.zeros(1, device=dev)
NCU = torch.cuda.get_device_properties(0).multi_processor_count


def masked_stream(pred):
    words = (NCU + 31) // 32
    mask = (C.c_uint32 * words)()
    for i in range(NCU):
        if pred(i):
            mask[i // 32] |= 1 << (i % 32)
    s = C.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(C.byref(s), words, mask)
    assert rc == 0, rc
    return torch.cuda.ExternalStream(s.value, device=dev), s


def where(stream, nwg=2048, threads=256, lds=0):
    out = torch.zeros(nwg, dtype=torch.int32, device=dev)
    rc = probe.xcc_probe(C.c_void_p(out.data_ptr()), nwg, threads, lds, 20000,
                         C.c_void_p(stream.cuda_stream))
    assert rc == 0, rc
    torch.cuda.synchronize()
    v = out.cpu().numpy().astype("uint32")
    xcc = collections.Counter(int(x >> 24) for x in v)
    cus = len({(int(x >> 24), int((x >> 13) & 7), int((x >> 12) & 1), int((x >> 8) & 15)) for x in v})
    return dict(sorted(xcc.items())), cus


print("CUs", NCU)
print("unmasked:", where(torch.cuda.current_stream(dev)))
for name, pred in (("bits i%8==7", lambda i: i % 8 == 7), ("bits i%8!=7", lambda i: i % 8 != 7),
                   ("bits 224..255", lambda i: i >= 224), ("bits 0..223", lambda i: i < 224)):
    st, keep = masked_stream(pred)
    print(f"{name:14s}:", where(st), " big-LDS 1024-thread WGs:", where(st, 512, 1024, 150 * 1024))
