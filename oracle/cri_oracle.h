/* oracle/cri_oracle.h -- TEST INFRASTRUCTURE ONLY.
 *
 * Plain-C CPU restatement of the reference's ADX / HCA algorithms (Youjose/PyCriCodecs @ 2024_08_07).
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library, and only as
 * the checker.  The product (pycricodecs_amd/, include/) never includes, links or calls anything in oracle/.
 *
 * Parity status: PINNED.  Every entry point is checked byte-for-byte against the real reference compiled
 * from /root/reference (oracle/_ref/criref, see oracle/Makefile + oracle/ref_harness.cpp) by
 * tests/test_oracle_vs_reference.py (in the build container) and against the committed golden vectors under
 * tests/golden/ (everywhere).
 *
 * Semantics = "reference with zero-initialised buffers and in-bounds accesses" (SURVEY.md section 9).
 * Return values use the same code space as include/cricodecs_hip.h.
 */
#ifndef CRI_ORACLE_H
#define CRI_ORACLE_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* ADX: adx.cpp:380-415 (decode), 416-506 (encode). */
int ora_adx_decode(const uint8_t* adx, size_t len, uint8_t** out, size_t* out_len);
int ora_adx_encode(const uint8_t* wav, size_t len, uint32_t bitdepth, uint32_t blocksize, uint32_t mode,
                   uint32_t highpass, uint32_t filter, uint32_t version, int force_no_loop,
                   uint8_t** out, size_t* out_len);

/* HCA: hca.cpp:3340-3457 (decode driver), 3459-3489 (encode driver), 3271-3337 (crypt). */
int ora_hca_decode(const uint8_t* hca, size_t len, uint64_t key, uint16_t subkey, uint8_t** out, size_t* out_len);
/* Same decode, but returns the pre-clamp float PCM of every frame: [frame][1024][channels] floats. */
int ora_hca_decode_float(const uint8_t* hca, size_t len, uint64_t key, uint16_t subkey, float** out, size_t* out_count);
int ora_hca_encode(const uint8_t* wav, size_t len, uint32_t force_no_loop, uint32_t quality,
                   uint8_t** out, size_t* out_len);
int ora_hca_crypt(uint8_t* hca, size_t len, uint32_t encrypt, uint32_t type, uint64_t key, uint16_t subkey);

/* Small pieces exposed for unit tests. */
uint16_t ora_crc16(const uint8_t* p, size_t n);                       /* hca.cpp:205-211 */
int ora_cipher_table(uint32_t type, uint64_t key, uint8_t table[256]); /* hca.cpp:499-617 */
void ora_adx_coefficients(uint32_t highpass, uint32_t rate, int32_t coef[2]); /* adx.cpp:58-64 */

void ora_free(void* p);

#ifdef __cplusplus
}
#endif
#endif
