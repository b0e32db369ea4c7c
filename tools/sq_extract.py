"""profiles/r06_record_loop_sq.json (tools/loop_sq.sh) -> profiles/r06_c3_fused_wave_sq.json: per-wave instruction counts, the
shares of a resident wave's time, per-SIMD unit occupancy of the record-loop kernels.  usage: python tools/sq_extract.py [tag]"""
import json, os, sys
R = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'profiles')
tag = sys.argv[1] if len(sys.argv) > 1 else 'r06'
d = json.load(open(os.path.join(R, '%s_record_loop_sq.json' % tag)))


def ext(cfg, k):
    v = d[cfg][k]
    w, cyc = v['SQ_WAVES'], v['GRBM_GUI_ACTIVE'] / 8.0
    keep = ('SQ_WAVES', 'SQ_BUSY_CYCLES', 'SQ_WAVE_CYCLES', 'SQ_WAIT_ANY', 'SQ_WAIT_INST_ANY', 'SQ_ACTIVE_INST_ANY', 'SQ_INSTS_VALU',
            'SQ_INSTS_SALU', 'SQ_INSTS_VMEM_RD', 'SQ_INSTS_VMEM_WR', 'SQ_INSTS_LDS', 'SQ_INSTS_BRANCH', 'SQ_ACTIVE_INST_VALU',
            'SQ_ACTIVE_INST_SCA', 'SQ_ACTIVE_INST_LDS', 'GRBM_GUI_ACTIVE', 'TCC_HIT_sum', 'TCC_MISS_sum', 'TCP_PENDING_STALL_CYCLES_sum')
    return {'counters': {c: v[c] for c in keep if c in v},
            'per_wave': {'valu': round(v['SQ_INSTS_VALU'] / w), 'salu': round(v['SQ_INSTS_SALU'] / w),
                         'vmem_rd': round(v['SQ_INSTS_VMEM_RD'] / w, 1), 'vmem_wr': round(v['SQ_INSTS_VMEM_WR'] / w, 1),
                         'lds': round(v['SQ_INSTS_LDS'] / w, 1), 'branch': round(v['SQ_INSTS_BRANCH'] / w, 1)},
            'wave_time_shares': v['derived'], 'kernel_cycles_per_xcd': round(cyc),
            'simd_valu_busy': round(v['SQ_ACTIVE_INST_VALU'] * 4 / 1024 / cyc, 3),
            'scalar_busy_per_simd': round(v['SQ_ACTIVE_INST_SCA'] * 4 / 1024 / cyc, 3),
            'waves_in_flight_per_simd': round(v['SQ_WAVE_CYCLES'] * 4 / 1024 / cyc, 2),
            'l2_hit_rate': round(v['TCC_HIT_sum'] / (v['TCC_HIT_sum'] + v['TCC_MISS_sum']), 3)}


def pick(cfg, stem):
    names = [k for k in d[cfg] if k.startswith(stem)]
    return max(names, key=lambda k: d[cfg][k].get('dispatches', 0))


out = {'_about': d['_about'] + '.  Extract (tools/sq_extract.py): per-wave instruction counts, the share of a wave\'s resident '
       'quad-cycles spent parked (s_waitcnt), issue-stalled and issuing, and per-SIMD unit occupancy = SQ_ACTIVE_INST_<unit> '
       'quad-cycles x 4 / 1024 SIMDs / kernel cycles (GRBM_GUI_ACTIVE / 8 XCDs).  One MI355X; the full set is '
       '%s_record_loop_sq.json.' % tag}
fw, sk = pick('C3', 'fused_wave_kernel<true'), pick('C2', 'stream_kernel')
out['C3 ' + fw] = ext('C3', fw)
out['C2 ' + sk] = ext('C2', sk)
out['C2 ordered_kernel'] = ext('C2', 'ordered_kernel')
out['C3 rl_place_kernel'] = ext('C3', 'rl_place_kernel')
f, s_ = out['C3 ' + fw], out['C2 ' + sk]
out['reading'] = (
    '%s: %d single-wave workgroups, %.1f waves in flight per SIMD on average of the 4 its 128 VGPRs allow; a wave issues %d vector + %d '
    'scalar instructions for its 16 384 records (%d vector lane-instructions per record) beside %.0f loads: the vector unit of a SIMD is '
    'busy %.0f %% of the kernel, the scalar unit %.0f %%; a resident wave spends %.0f %% of its time parked on s_waitcnt, %.0f %% ready but '
    'not issued, %.0f %% issuing.  %s (C2): %d vector instructions per wave, parked %.0f %% - a streaming kernel.  The headline loop is '
    'bound by memory AND issue together: it moves its bytes at 0.93-0.97 of the box\'s copy rate, and bounded to three waves per SIMD it '
    'runs as fast as with four (tools/ab.sh): more waves in flight would not help.'
    % (fw, f['counters']['SQ_WAVES'], f['waves_in_flight_per_simd'], f['per_wave']['valu'], f['per_wave']['salu'],
       round(f['per_wave']['valu'] * 64 / 16384.0), f['per_wave']['vmem_rd'], 100 * f['simd_valu_busy'], 100 * f['scalar_busy_per_simd'],
       100 * f['wave_time_shares']['wait_any_of_wave_cycles'], 100 * f['wave_time_shares']['wait_inst_of_wave_cycles'],
       100 * f['wave_time_shares']['active_inst_of_wave_cycles'], sk, s_['per_wave']['valu'],
       100 * s_['wave_time_shares']['wait_any_of_wave_cycles']))
json.dump(out, open(os.path.join(R, '%s_c3_fused_wave_sq.json' % tag), 'w'), indent=1)
print(out['reading'])
