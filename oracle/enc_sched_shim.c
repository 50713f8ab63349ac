/* enc_sched_shim.c — test infrastructure, linked into the two encoder applications oracle/Makefile.enc builds (never into the product).
 *
 * The reference switches the thread that initialises the encoder to SCHED_FIFO priority 99 (enc_switch_to_real_time,
 * Source/Lib/Encoder/Globals/EbEncHandle.c:296-305) and creates every pipeline thread with SCHED_FIFO priority 99 and PTHREAD_EXPLICIT_SCHED
 * (Source/Lib/Common/Codec/EbThreads.c:85-93); it falls back to the default policy only when that is refused (EPERM, i.e. when it does not
 * run as root).  The test containers do run as
 * root, and on a shared VM whose real-time bandwidth is throttled ("sched: RT throttling activated") those threads are runnable but never
 * scheduled: the application then polls svt_av1_enc_get_packet forever.  The scheduling policy has no influence on the bitstream, so the
 * harness answers both requests with success and changes nothing: the attribute stays at PTHREAD_INHERIT_SCHED and the threads inherit the
 * default policy of the main thread — the state the reference's own EPERM path ends in.
 */
#include <pthread.h>
#include <sched.h>

int pthread_attr_setinheritsched(pthread_attr_t *attr, int inherit) {
    (void)attr;
    (void)inherit;
    return 0;
}

int pthread_setschedparam(pthread_t thread, int policy, const struct sched_param *param) {
    (void)thread;
    (void)policy;
    (void)param;
    return 0;
}
