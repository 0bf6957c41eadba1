/* oracle/ref_shim.h -- TEST INFRASTRUCTURE ONLY.
 *
 * Force-included (g++ -include) in front of the UNMODIFIED reference source
 * /root/reference/src/pbsim.cpp to build oracle/_ref/pbsim_ref_philox: every
 * `rand()` call site of the reference (111 sites, pbsim.cpp v3.0.5) is routed
 * to pbshim_draw(), which serves the keyed Philox stream of DESIGN.md "RNG
 * contract" instead of glibc's sequential stream.  The reference's own code
 * therefore generates the Philox-mode golden outputs; nothing of the reference
 * is copied or edited.
 *
 * The macro names identifiers that exist at every walk call site of the
 * reference: `h` (pass index), `maf_offset` (MAF column = HMM event index),
 * `sim.res_num` (read counter) and `genome.num` (FASTA record number, 0 for
 * trans/templ).  The two file-scope fallbacks below only make the macro
 * compile inside simulate_by_sample(), which has no `h`.
 */
#ifndef PBSIM_ORACLE_REF_SHIM_H
#define PBSIM_ORACLE_REF_SHIM_H
#include <stdlib.h>
static long h = 0, maf_offset = 0;
extern "C" int pbshim_draw(int line, long unit, long read, long pass, long event);
#define rand() pbshim_draw(__LINE__, (long)genome.num, (long)sim.res_num, (long)h, (long)maf_offset)
#endif
