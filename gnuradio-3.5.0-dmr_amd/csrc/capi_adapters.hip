// capi_adapters.hip -- the remaining harness adapters of SURVEY 8f n4: gr_stream_to_vector, gr_head
// (gr_vector_to_streams moves data exactly like gr_stream_to_streams and shares grhip_stream_adapter_*).
//   general/gr_stream_to_vector.cc:46-60  work = memcpy(out, in, noutput_items * item_size * nitems_per_block)
//   general/gr_head.cc:44-62              copies until nitems have passed, then returns -1 (WORK_DONE)
// Both are plain copies: on device pointers a hipMemcpyAsync on the caller's stream, no kernel of our own.
#include "grhip_internal.h"

using namespace grhip;

struct grhip_copy_adapter : HandleBase {
    size_t item_size = 1;                 // bytes per OUTPUT item
    bool head = false;
    unsigned long long nitems = 0, ncopied = 0;
};

static int adapter_create(grhip_copy_adapter **h, size_t item_bytes, bool head, unsigned long long nitems, int device)
{
    if (!h) return fail(GRHIP_EINVAL, "null argument");
    *h = nullptr;
    if (item_bytes == 0) return fail(GRHIP_EINVAL, "item size must be > 0");
    auto *b = new (std::nothrow) grhip_copy_adapter();
    if (!b) return fail(GRHIP_ENOMEM, "alloc");
    b->item_size = item_bytes; b->head = head; b->nitems = nitems;
    int rc = b->init_device(device);
    if (rc) { b->destroy_base(); delete b; return rc; }
    *h = b;
    return GRHIP_OK;
}

extern "C" {

int grhip_stream_to_vector_create(grhip_copy_adapter **h, size_t item_size, size_t nitems_per_block, int device)
{
    if (nitems_per_block == 0) return fail(GRHIP_EINVAL, "nitems_per_block must be > 0");
    return adapter_create(h, item_size * nitems_per_block, false, 0, device);
}

int grhip_head_create(grhip_copy_adapter **h, size_t sizeof_stream_item, unsigned long long nitems, int device)
{
    return adapter_create(h, sizeof_stream_item, true, nitems, device);
}

void grhip_copy_adapter_destroy(grhip_copy_adapter *h)
{
    if (!h) return;
    h->destroy_base();
    delete h;
}

int grhip_head_reset(grhip_copy_adapter *h)           // gr_head::reset(), general/gr_head.h:51
{
    if (!h) return fail(GRHIP_EINVAL, "null handle");
    h->ncopied = 0;
    return GRHIP_OK;
}

// in / out: host pointers (device = 0) or device pointers (device = 1, copy queued on `stream`)
static int adapter_work(grhip_copy_adapter *h, int noutput_items, const void *in, void *out, bool on_device, void *stream)
{
    if (!h) return fail(GRHIP_EINVAL, "null handle");
    if (noutput_items < 0) return fail(GRHIP_EINVAL, "negative item count");
    unsigned long long n = (unsigned long long)noutput_items;
    if (h->head) {
        if (h->ncopied >= h->nitems) return GRHIP_WORK_DONE;                      // gr_head.cc:49-50: done (-1 there)
        n = std::min(h->nitems - h->ncopied, n);                                  // .cc:52
        n = std::min(n, (unsigned long long)(GRHIP_WORK_DONE - 1));               // an item count is never the sentinel
    }
    if (n == 0) return 0;
    if (!in || !out) return fail(GRHIP_EINVAL, "null buffer");
    if (on_device) {
        int rc = h->bind();
        if (rc) return rc;
        GRHIP_HIP(hipMemcpyAsync(out, in, (size_t)n * h->item_size, hipMemcpyDeviceToDevice, h->pick(stream)));
    } else {
        memcpy(out, in, (size_t)n * h->item_size);
    }
    h->ncopied += n;
    return (int)n;
}

int grhip_copy_adapter_work(grhip_copy_adapter *h, int noutput_items, const void *in, void *out)
{
    return adapter_work(h, noutput_items, in, out, false, nullptr);
}

int grhip_copy_adapter_work_device(grhip_copy_adapter *h, int noutput_items, const void *d_in, void *d_out, void *stream)
{
    return adapter_work(h, noutput_items, d_in, d_out, true, stream);
}

}  // extern "C"
