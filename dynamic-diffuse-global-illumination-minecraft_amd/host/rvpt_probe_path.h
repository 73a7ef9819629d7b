// rvpt_probe_path.h — C++ host-side mirror of the probe path of the reference's `class RVPT`
// (src/rvpt/rvpt.h:33-92), header-only, over the C ABI of include/ddgi_probe.h.
//
// Same member names, argument meaning and call order as the reference for this path:
//   RVPT rvpt(window);                 ->  RVPTProbePath rvpt;
//   rvpt.generate_probe_rays();            rvpt.generate_probe_rays();      (main.cpp:47)
//   rvpt.initialize();                     rvpt.initialize();               (main.cpp:48)
//   loop: rvpt.update(); rvpt.draw();      rvpt.update(); rvpt.draw();      (main.cpp:93-95)
//   rvpt.shutdown();                       rvpt.shutdown();
// `ir` and `render_settings` are the public data members the UI code of the reference writes
// (rvpt.cpp:327-364); recreate_probe_textures() is the "Recalculate Probes" button (rvpt.cpp:361).
// Error behaviour: like the reference's initialize()/update() the calls return bool and print the
// reason to stderr (rvpt.cpp:499-592); nothing throws.
#pragma once

#include <cstdio>
#include <vector>

#include "../../include/ddgi_probe.h"

class RVPTProbePath
{
public:
    // RVPT::RenderSettings / RVPT::IrradianceField with the reference's defaults (rvpt.h:72-89)
    ddgi_render_settings render_settings{1600, 900, 8, 0, 0, 0, 0.f, 0};
    ddgi_irradiance_field ir{{9, 7, 9}, 11, 0.9f, 20, {0, 0}, {1.4f, 0.f, 1.f}, 1, {0, 0, 0}};

    explicit RVPTProbePath(int device = 0) : device_(device) {}
    // One z-slab of a probe grid sharded over `world` GPUs (the reference is single-GPU): this object
    // traces + blends the probes with z in [rank*cz/world, (rank+1)*cz/world) on `device`.  `nccl_comm` is an
    // ncclComm_t of that size/rank (ddgi_comm_create / ddgi_comm_create_all, or the host's own RCCL); draw() then
    // also issues the in-place all-gather of the probe textures, pipelined behind the next frame's update.
    RVPTProbePath(int device, int rank, int world, void* nccl_comm) : device_(device), rank_(rank), world_(world), comm_(nccl_comm) {}
    // The same slab with the peer-to-peer exchange instead of RCCL (one process per rank; ddgi_exchange_p2p_*): every rank pushes
    // its slab into its peers' IPC-mapped textures.  `gather(mine, all, bytes, user)` is the host's all-gather of `bytes` per rank
    // in rank order (MPI_Allgather, a socket, pipes through a parent ...) — the only thing the library cannot do by itself; it
    // returns false on failure.  Called by initialize() and again whenever the textures are recreated.
    using AddressGather = bool (*)(const uint8_t* mine, uint8_t* all, size_t bytes, void* user);
    RVPTProbePath(int device, int rank, int world, AddressGather gather, void* user) : device_(device), rank_(rank), world_(world), gather_(gather), gather_user_(user) {}
    ~RVPTProbePath() { shutdown(); }
    RVPTProbePath(const RVPTProbePath&) = delete;
    RVPTProbePath& operator=(const RVPTProbePath&) = delete;

    // rvpt.cpp:1177-1224.  Before initialize() it only remembers that rays are wanted (the
    // reference sizes its SSBO from this call, main.cpp:47); afterwards it regenerates + uploads.
    void generate_probe_rays()
    {
        need_generate_probe_rays_ = true;
        if (handle_) (void)flush_rays();
    }

    // rvpt.cpp:231-264 (probe resources only)
    bool initialize()
    {
        if (handle_) return true;
        if (!ok(ddgi_create_sharded(&ir, &render_settings, device_, rank_, world_, &handle_), "ddgi_create_sharded")) return false;
        if (!attach_exchange()) return false;
        return flush_rays();
    }

    // rvpt.cpp:265-290: advance time, make the per-frame uploads current
    bool update()
    {
        render_settings.time += 2;  // rvpt.cpp:281
        if (need_change_probe_texture_) return recreate_probe_textures();
        return flush_rays();
    }

    // probe half of rvpt.cpp:372-431 (record_compute_command_buffer 1096-1129 + submit)
    // Sharded: + the all-gather of this frame's textures (asynchronous; the sample / read calls below wait for it).
    // A process that drives several slabs brackets the draw() calls of one frame with ddgi_exchange_group_begin/end.
    bool draw()
    {
        if (!handle_ || !ok(ddgi_probe_update(handle_, &render_settings), "ddgi_probe_update")) return false;
        return !(comm_ || gather_) || ok(ddgi_exchange(handle_), "ddgi_exchange");
    }

    // rvpt.cpp:661-755; keep_surviving_probes: probes that stand where an old probe stood keep their tiles
    // (ddgi_reconfigure) — the reference itself drops every texture
    bool recreate_probe_textures(bool keep_surviving_probes = false)
    {
        need_change_probe_texture_ = false;
        if (!handle_) return initialize();
        if (!ok(ddgi_reconfigure(handle_, &ir, &render_settings, keep_surviving_probes ? 1 : 0), "ddgi_reconfigure")) return false;
        if (!attach_exchange()) return false;  // reconfiguring detaches the exchange
        need_generate_probe_rays_ = true;
        return flush_rays();
    }
    void request_probe_texture_change() { need_change_probe_texture_ = true; }  // the UI sliders, rvpt.cpp:334-357

    // rvpt.cpp:433-468
    void shutdown()
    {
        if (handle_) (void)ddgi_destroy(handle_);
        handle_ = nullptr;
    }

    // what the render pass reads through bindings 3/4 (rvpt.cpp:783-786), here copied to the host
    bool read_probe_textures(std::vector<uint8_t>& albedo, std::vector<uint8_t>& distance)
    {
        int w = 0, h = 0;
        ddgi_texture_size(&ir, &w, &h);
        albedo.resize(static_cast<size_t>(w) * h * 4);
        distance.resize(albedo.size());
        return handle_ && ok(ddgi_read_textures(handle_, albedo.data(), distance.data()), "ddgi_read_textures");
    }

    // get_diffuse_gi for a batch of shading points (intersection.glsl:1306-1409)
    bool sample(const float* pos_xyz, const float* nrm_xyz, size_t n, float* rgb_out, int32_t* cage_idx8_out = nullptr)
    {
        return handle_ && ok(ddgi_sample(handle_, pos_xyz, nrm_xyz, n, rgb_out, cage_idx8_out), "ddgi_sample");
    }

    bool wait() { return handle_ && ok(ddgi_synchronize(handle_), "ddgi_synchronize"); }
    ddgi_handle native_handle() const { return handle_; }

private:
    // pipelined in both transports: two texture pairs, the exchange of frame k behind the update of frame k + 1
    bool attach_exchange()
    {
        if (comm_) return ok(ddgi_exchange_init(handle_, comm_, 1), "ddgi_exchange_init");
        if (!gather_) return true;
        uint8_t mine[DDGI_P2P_ADDRESS_BYTES];
        if (!ok(ddgi_exchange_p2p_export(handle_, 1, mine), "ddgi_exchange_p2p_export")) return false;
        std::vector<uint8_t> all(static_cast<size_t>(world_) * DDGI_P2P_ADDRESS_BYTES);
        if (!gather_(mine, all.data(), DDGI_P2P_ADDRESS_BYTES, gather_user_))
        {
            std::fprintf(stderr, "the host's address all-gather failed\n");
            return false;
        }
        // The ranks map their peers' buffers ONE AFTER THE OTHER (the same all-gather, one byte, as the barrier between the turns): a rank inside
        // hipIpcOpenMemHandle waits for the EXPORTING process to hand a buffer over, and on the stack measured two processes that attach to each
        // other at the same time can wait for each other forever (docs/LAB_NOTES.md "Round 6"); a turn costs milliseconds.
        bool good = true;
        std::vector<uint8_t> turn_done(static_cast<size_t>(world_));
        for (int turn = 0; turn < world_; ++turn)
        {
            if (turn == rank_) good = ok(ddgi_exchange_p2p_init(handle_, all.data(), world_), "ddgi_exchange_p2p_init");
            const uint8_t mine_done = good ? 1 : 0;
            if (!gather_(&mine_done, turn_done.data(), 1, gather_user_))
            {
                std::fprintf(stderr, "the host's all-gather failed (turn %d of the peer mapping)\n", turn);
                return false;
            }
            if (!turn_done[static_cast<size_t>(turn)]) return false;  // (that rank could not map its peers: every rank gives the transport up)
        }
        return good;
    }
    bool flush_rays()
    {
        if (!need_generate_probe_rays_) return true;
        need_generate_probe_rays_ = false;
        return ok(ddgi_generate_probe_rays(handle_, 1u, 0), "ddgi_generate_probe_rays");
    }
    static bool ok(int rc, const char* what)
    {
        if (rc != DDGI_OK) std::fprintf(stderr, "%s failed (%d): %s\n", what, rc, ddgi_last_error());
        return rc == DDGI_OK;
    }

    int device_ = 0, rank_ = 0, world_ = 1;
    void* comm_ = nullptr;  // ncclComm_t, caller-owned
    AddressGather gather_ = nullptr;  // peer-to-peer exchange: the host's all-gather of the 512-byte addresses
    void* gather_user_ = nullptr;
    ddgi_handle handle_ = nullptr;
    bool need_generate_probe_rays_ = true;    // rvpt.h:96
    bool need_change_probe_texture_ = false;  // rvpt.h:95
};
