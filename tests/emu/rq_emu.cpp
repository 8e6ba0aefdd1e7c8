// rq_emu.cpp -- TEST INFRASTRUCTURE ONLY (see rq_emu.h).  Fiber scheduler for emulated workgroups.
#include "rq_emu.h"

namespace rqemu {

ucontext_t g_main;
Fiber* g_cur = nullptr;
Block* g_blk = nullptr;
dim3 g_blockIdx, g_blockDim, g_gridDim;
std::function<void()> g_entry;

static std::vector<char*> g_stacks;
bool g_dma_late = false;

unsigned char* dyn_smem() { return g_blk->dyn.data(); }
size_t dyn_smem_size() { return g_blk->dyn.size(); }

static void trampoline() {
    g_entry();
    Fiber& f = *g_cur;
    dma_land_until(f, 0);          // a wavefront's outstanding DMAs complete before it retires
    f.done = true;
    Block& b = *g_blk;
    b.waves[f.wave].nalive--;
    b.nalive--;
    b.progress++;
    if (b.nalive > 0 && b.bar_arrived >= b.nalive) { b.bar_arrived = 0; b.bar_gen++; }
    swapcontext(&f.ctx, &g_main);
}

void run_grid(dim3 grid, dim3 block, size_t smem) {
    const int nthreads = block.x * block.y * block.z;
    const int nwaves = (nthreads + kWave - 1) / kWave;
    while ((int)g_stacks.size() < nthreads) g_stacks.push_back((char*)malloc(kStack));
    g_blockDim = block;
    g_gridDim = grid;
    {
        const char* e = getenv("RQ_EMU_DMA");
        g_dma_late = e && strcmp(e, "late") == 0;
    }
    for (unsigned bz = 0; bz < grid.z; ++bz)
        for (unsigned by = 0; by < grid.y; ++by)
            for (unsigned bx = 0; bx < grid.x; ++bx) {
                Block blk;
                g_blk = &blk;
                g_blockIdx = dim3(bx, by, bz);
                blk.fibers.resize(nthreads);
                blk.waves.resize(nwaves);
                blk.dyn.assign(smem + 64, 0xFF);   // NaN-ish poison: uninitialised LDS reads show up
                blk.nalive = nthreads;
                for (int t = 0; t < nthreads; ++t) {
                    Fiber& f = blk.fibers[t];
                    f.tid = dim3(t % block.x, (t / block.x) % block.y, t / (block.x * block.y));
                    f.lane = t % kWave;
                    f.wave = t / kWave;
                    blk.waves[f.wave].nalive++;
                    f.stack = g_stacks[t];
                    getcontext(&f.ctx);
                    f.ctx.uc_stack.ss_sp = f.stack;
                    f.ctx.uc_stack.ss_size = kStack;
                    f.ctx.uc_link = nullptr;
                    makecontext(&f.ctx, trampoline, 0);
                }
                int remaining = nthreads;
                while (remaining > 0) {
                    unsigned long before = blk.progress;
                    int ran = 0;
                    for (int t = 0; t < nthreads; ++t) {
                        Fiber& f = blk.fibers[t];
                        if (f.done) continue;
                        g_cur = &f;
                        swapcontext(&g_main, &f.ctx);
                        ++ran;
                        if (f.done) --remaining;
                    }
                    if (remaining > 0 && blk.progress == before) {
                        fprintf(stderr, "rqemu: deadlock in block (%u,%u,%u): %d fibers stuck "
                                        "(divergent barrier or collective)\n", bx, by, bz, remaining);
                        abort();
                    }
                }
            }
    g_blk = nullptr;
    g_cur = nullptr;
}

}  // namespace rqemu
