"""The occupancy the design rests on, checked on the compiler's own report of the build (projectd-core_amd/kernel_resources.txt,
written by the product Makefile from -Rpass-analysis=kernel-resource-usage): six workgroups of the first-pass kernel per CU need
at most 80 VGPRs a lane (512 / 6, granule 8) and at most 26880 bytes of LDS a workgroup (160 KB in 1280-byte granules: 21 of
them) -- 26888 bytes once cost the sixth workgroup and 12 % of the headline rate without any test noticing."""
import os, re
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
REPORT = os.path.join(HERE, '..', 'projectd-core_amd', 'kernel_resources.txt')


def kernels():
    txt = open(REPORT).read()
    out = {}
    for m in re.finditer(r'Function Name: (\w+)(.*?)(?=Function Name:|\Z)', txt, re.S):
        body = m.group(2)
        g = lambda key: int(re.search(key + r': (\d+)', body).group(1))
        out[m.group(1)] = dict(vgprs=g('VGPRs'), lds=g(r'LDS Size \[bytes/block\]'), scratch=g(r'ScratchSize \[bytes/lane\]'), occupancy=g(r'Occupancy \[waves/SIMD\]'))
    return out


@pytest.mark.skipif(not os.path.exists(REPORT), reason='product library not built by its Makefile here')
def test_first_pass_kernels_keep_six_workgroups_per_cu():
    k = kernels()
    for name in ('pdb_step_kernel', 'pdb_step_kernel_generic'):
        assert k[name]['vgprs'] <= 80, (name, k[name])
        assert k[name]['lds'] <= 26880, (name, k[name])
        assert k[name]['occupancy'] >= 6, (name, k[name])
    # the contact pass holds few workgroups (the cars that touch something): it trades occupancy for registers and LDS -- two
    # workgroups per CU (256 VGPRs, no vector spills worth the name) and the collision staging block in LDS (64 KB is the limit of a
    # statically allocated workgroup)
    for name in ('pdb_contact_kernel', 'pdb_contact_kernel_generic', 'pdb_contact_kernel_wide', 'pdb_contact_kernel_ctrl'):
        assert k[name]['vgprs'] <= 256 and k[name]['occupancy'] >= 2, (name, k[name])
        assert k[name]['lds'] <= 65536 and k[name]['scratch'] <= 128, (name, k[name])
    # scratch of the first pass: the cold teleport block's frame (round 2: 72 bytes).  A hot-path spill shows up as a larger frame first.
    for name in ('pdb_step_kernel', 'pdb_step_kernel_generic'):
        assert k[name]['scratch'] <= 96, (name, k[name])
    for name in ('pdb_step_kernel_wide', 'pdb_step_kernel_ctrl'):   # 40-row cars: five workgroups (96 VGPRs, 29.9 KB)
        assert k[name]['vgprs'] <= 96 and k[name]['lds'] <= 32000, (name, k[name])
