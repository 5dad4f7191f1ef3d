"""Device-wide hardware counters over the regime bench.py's `value` is measured in (engine.Pipeline, D batches in flight, replayed graphs).
  ROCP_TOOL_LIBRARIES=$PWD/tools/devcount/libdevcount.so python tools/inflight_counters.py [depth=4] [steps=400] [out.json]
One counter set per timed region (the SQ block has 8 slots, TCC 4, GRBM 2); every region is the same K steps, so the sets can be read side by
side.  Prints / writes {set: {counter: {sum, n, max}}, us_per_step, images_per_sec} per set, and derived chip-wide utilisations."""
import ctypes as C, json, os, sys, time
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')
import torch
from k210_yolo_framework_amd import engine, netspec
from k210_yolo_framework_amd.helper import VOC_ANCHORS

depth = int(sys.argv[1]) if len(sys.argv) > 1 else 4
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 400
out_path = sys.argv[3] if len(sys.argv) > 3 else os.path.join(root, 'gpurun_out', f'inflight_counters_d{depth}.json')
B = int(os.environ.get('YK_BENCH_BATCH', '32'))
dc = C.CDLL(os.path.join(root, 'tools', 'devcount', 'libdevcount.so'))
dc.devcount_start.argtypes = [C.c_char_p]
dc.devcount_stop.argtypes = [C.c_char_p, C.c_int]

SETS = {
    'waves': 'GRBM_GUI_ACTIVE,GRBM_COUNT,SQ_WAVES,SQ_BUSY_CYCLES,SQ_BUSY_CU_CYCLES,SQ_WAVE_CYCLES,SQ_WAIT_ANY,SQ_WAIT_INST_ANY,SQ_ACTIVE_INST_ANY,SQ_LEVEL_WAVES',
    'issue': 'GRBM_GUI_ACTIVE,SQ_WAVE_CYCLES,SQ_ACTIVE_INST_VALU,SQ_ACTIVE_INST_LDS,SQ_ACTIVE_INST_VMEM,SQ_ACTIVE_INST_SCA,SQ_ACTIVE_INST_MISC,SQ_VALU_MFMA_BUSY_CYCLES,SQ_INST_CYCLES_VMEM_RD',
    'insts': 'GRBM_GUI_ACTIVE,SQ_INSTS_VALU,SQ_INSTS_MFMA,SQ_INSTS_LDS,SQ_INSTS_VMEM_RD,SQ_INSTS_VMEM_WR,SQ_INSTS_SALU,SQ_INSTS_SMEM,SQ_THREAD_CYCLES_VALU',
    'lds': 'GRBM_GUI_ACTIVE,SQ_LDS_IDX_ACTIVE,SQ_LDS_BANK_CONFLICT,SQ_LDS_ADDR_CONFLICT,SQ_WAIT_INST_LDS,SQ_LDS_DATA_FIFO_FULL,SQ_LDS_CMD_FIFO_FULL,SQ_VMEM_TA_ADDR_FIFO_FULL,SQ_VMEM_TA_CMD_FIFO_FULL',
    'spi_a': 'GRBM_GUI_ACTIVE,SPI_RA_REQ_NO_ALLOC_CSN,SPI_RA_RES_STALL_CSN',
    'spi_b': 'GRBM_GUI_ACTIVE,SPI_RA_LDS_CU_FULL_CSN,SPI_RA_VGPR_SIMD_FULL_CSN',
    'spi_c': 'GRBM_GUI_ACTIVE,SPI_RA_WAVE_SIMD_FULL_CSN,SPI_RA_SGPR_SIMD_FULL_CSN',
    'spi_d': 'GRBM_GUI_ACTIVE,SPI_RA_BAR_CU_FULL_CSN,SPI_RA_TGLIM_CU_FULL_CSN',
    'spi_e': 'GRBM_GUI_ACTIVE,SPI_CSN_BUSY,SPI_CSN_NUM_THREADGROUPS',
    'spi_f': 'GRBM_GUI_ACTIVE,SPI_CSN_WAVE,SPI_CSN_WINDOW_VALID',
    'coexec': 'GRBM_GUI_ACTIVE,SQ_VALU_MFMA_COEXEC_CYCLES,SQ_VALU_MFMA_BUSY_CYCLES,SQ_ACTIVE_INST_VALU,SQ_INSTS_VALU_CVT,SQ_INSTS_VALU_FMA_F32,SQ_INSTS_VALU_INT32,SQ_INSTS_VALU_MUL_F32,SQ_INSTS_VALU_ADD_F32',
    'l2': 'GRBM_GUI_ACTIVE,TCC_REQ_sum,TCC_HIT_sum,TCC_MISS_sum,TCC_EA0_RDREQ_sum',
    'hbm': 'GRBM_GUI_ACTIVE,TCC_EA0_RDREQ_32B_sum,TCC_EA0_WRREQ_sum,TCC_EA0_WRREQ_64B_sum,TCC_BUSY_sum',
    'tcp': 'GRBM_GUI_ACTIVE,TCP_TOTAL_CACHE_ACCESSES_sum,TCP_TCC_READ_REQ_sum,TCP_TCC_WRITE_REQ_sum,TCP_PENDING_STALL_CYCLES_sum',
    'ta': 'GRBM_GUI_ACTIVE,TA_TA_BUSY_sum,TA_BUFFER_READ_LDS_WAVEFRONTS_sum',
    'ta2': 'GRBM_GUI_ACTIVE,TA_BUFFER_WAVEFRONTS_sum,TA_ADDR_STALLED_BY_TC_CYCLES_sum',
    'td': 'GRBM_GUI_ACTIVE,TD_TD_BUSY_sum,TD_TC_STALL_sum',
}
only = os.environ.get('DC_SETS')
if only:
    SETS = {k: v for k, v in SETS.items() if k in only.split(',')}

spec = netspec.yolo_mobilev1((224, 320, 3), 3, 20, alpha=0.75)
pipe = engine.Pipeline(spec, spec.init_weights(seed=1), VOC_ANCHORS, max_batch=B, depth=depth, precision='f16x2', graph=True)
frames = torch.randint(0, 256, (B, 224, 320, 3), dtype=torch.uint8, device='cuda', generator=torch.Generator(device='cuda').manual_seed(0))


def region(n):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        pipe.submit(frames, sync_input=False)
    torch.cuda.synchronize()
    return time.perf_counter() - t0


region(5 * depth)
plain = region(steps)
res = {'depth': depth, 'steps': steps, 'batch': B, 'plain_us_per_step': plain / steps * 1e6, 'plain_images_per_sec': B * steps / plain,
       'ready': int(dc.devcount_ready()), 'sets': {}}
print(f'depth {depth}: plain {B * steps / plain:.0f} images/s, {plain / steps * 1e6:.1f} us/step; devcount ready={res["ready"]}', flush=True)
buf = C.create_string_buffer(1 << 16)
for name, ctrs in SETS.items():
    region(2 * depth)
    rc = dc.devcount_start(ctrs.encode())
    if rc != 0:
        res['sets'][name] = {'error': f'start rc={rc}'}
        print(name, 'start failed', rc, flush=True)
        continue
    el = region(steps)
    n = dc.devcount_stop(buf, len(buf))
    if n < 0:
        res['sets'][name] = {'error': f'stop rc={n}'}
        print(name, 'stop failed', n, flush=True)
        continue
    c = json.loads(buf.value.decode())
    res['sets'][name] = {'us_per_step': el / steps * 1e6, 'images_per_sec': B * steps / el, 'seconds': el, 'counters': c}
    print(name, f'{B * steps / el:.0f} images/s', {k: (v['sum'], v['n']) for k, v in c.items()}, flush=True)
json.dump(res, open(out_path, 'w'), indent=1)
pipe.close()
