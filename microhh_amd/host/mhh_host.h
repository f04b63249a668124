// mhh_host.h -- C++ host side of the drop-in: the reference's operator interfaces for the hot path, implemented by
// forwarding to the C ABI (include/mhh_hip.h).
//
// MicroHH selects its GPU backend at compile time: every operator's exec()/get_cfl()/... is defined once in the
// .cxx under `#ifndef USECUDA` and once in the .cu under `#ifdef USECUDA` (src/advec_2.cxx:265 <-> src/advec_2.cu:140,
// src/diff_smag2.cxx:882,902,937 <-> src/diff_smag2.cu:519,551,703,792,833, src/pres_2.cxx:64,97,389 <->
// src/pres_2.cu:215). This header provides those member functions -- same names, arguments and error behaviour
// (std::runtime_error on failure) -- over containers that mirror the slice of Grid_data / Field3d / Fields /
// Boundary / Thermo the operators touch. INTEGRATION.md shows the same bodies written against the reference's
// real classes (the adaptor translation units a MicroHH maintainer would add in place of the .cu files).
//
// Header-only, C++17, no HIP or torch types: device memory is owned by the caller (Field3d::init_device,
// src/field3d.cu:32-47) and travels as raw pointers.
#pragma once
#include <array>
#include <cmath>
#include <cstdio>
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>
#include "../../include/mhh_hip.h"

namespace mhh_host
{
template<typename TF> constexpr int mhh_dtype();
template<> constexpr int mhh_dtype<double>() { return MHH_F64; }
template<> constexpr int mhh_dtype<float>()  { return MHH_F32; }

inline void mhh_check(int rc) { if (rc != MHH_OK) throw std::runtime_error(std::string("mhh: ") + mhh_last_error()); }

// ---- containers (subset of include/grid.h:49-135, include/field3d.h:33-81, include/fields.h:132-161) ----------
template<typename TF>
struct Grid_data
{
    int itot, jtot, ktot, imax, jmax, kmax, igc, jgc, kgc;
    int icells, jcells, ijcells, kcells, ncells;
    int istart, jstart, kstart, iend, jend, kend;
    TF xsize, ysize, zsize, dx, dy;
    std::vector<TF> z, zh, dz, dzh, dzi, dzhi, dzi4, dzhi4;                    // host metrics [kcells]
    TF* z_g = nullptr; TF* zh_g = nullptr; TF* dz_g = nullptr; TF* dzh_g = nullptr;
    TF* dzi_g = nullptr; TF* dzhi_g = nullptr; TF* dzi4_g = nullptr; TF* dzhi4_g = nullptr;   // device copies (Grid::prepare_device)
    int npy = 1, mpicoordy = 0;                                                // slab decomposition (npx == 1)
};

template<typename TF>
struct Field3d
{
    TF* fld_g = nullptr;
    TF* flux_bot_g = nullptr; TF* flux_top_g = nullptr;
    TF visc = 0;
};

template<typename TF>
struct Fields
{
    using Map = std::map<std::string, std::shared_ptr<Field3d<TF>>>;
    Map mp, mt, sp, st, sd;            // momentum / tendencies / scalars / scalar tendencies / diagnostic (evisc, p)
    TF* rhoref_g = nullptr; TF* rhorefh_g = nullptr;
    std::vector<TF> rhoref, rhorefh;   // host copies (Pres::set_values runs on the host)
    TF visc = 0;
};

template<typename TF>
struct Grid
{
    Grid_data<TF> gd;
    const Grid_data<TF>& get_grid_data() const { return gd; }
    // C-ABI descriptor with device (default) or host metric pointers
    mhh_grid abi(bool host = false) const
    {
        mhh_grid g{};
        g.itot = gd.itot; g.jtot = gd.jtot; g.ktot = gd.ktot; g.imax = gd.imax; g.jmax = gd.jmax; g.kmax = gd.kmax;
        g.igc = gd.igc; g.jgc = gd.jgc; g.kgc = gd.kgc; g.icells = gd.icells; g.jcells = gd.jcells; g.ijcells = gd.ijcells; g.kcells = gd.kcells;
        g.istart = gd.istart; g.jstart = gd.jstart; g.kstart = gd.kstart; g.iend = gd.iend; g.jend = gd.jend; g.kend = gd.kend;
        g.dtype = mhh_dtype<TF>(); g.npx = 1; g.npy = gd.npy; g.mpicoordx = 0; g.mpicoordy = gd.mpicoordy;
        g.ncells = (long long)gd.ijcells * gd.kcells;
        g.xsize = gd.xsize; g.ysize = gd.ysize; g.zsize = gd.zsize; g.dx = gd.dx; g.dy = gd.dy;
        if (host) { g.z = gd.z.data(); g.zh = gd.zh.data(); g.dz = gd.dz.data(); g.dzh = gd.dzh.data(); g.dzi = gd.dzi.data(); g.dzhi = gd.dzhi.data(); g.dzi4 = gd.dzi4.data(); g.dzhi4 = gd.dzhi4.data(); }
        else      { g.z = gd.z_g; g.zh = gd.zh_g; g.dz = gd.dz_g; g.dzh = gd.dzh_g; g.dzi = gd.dzi_g; g.dzhi = gd.dzhi_g; g.dzi4 = gd.dzi4_g; g.dzhi4 = gd.dzhi4_g; }
        return g;
    }
};

// What Diff_smag2 reads from Boundary / Thermo (src/diff_smag2.cxx:1050-1180)
template<typename TF>
struct Boundary
{
    std::string swboundary = "default";
    TF* z0m_g = nullptr; TF* dudz_g = nullptr; TF* dvdz_g = nullptr; TF* dbdz_g = nullptr;
    std::string get_switch() const { return swboundary; }
};
template<typename TF>
struct Thermo
{
    std::string swthermo = "0";
    TF* N2_g = nullptr;            // get_thermo_field("N2") result, or null to have it evaluated from scalar `th`
    std::string th = "th"; TF* thref_g = nullptr; TF* threfh_g = nullptr; TF grav = 9.81;
    int swspatialorder = 2;        // grid.swspatialorder: calc_buoyancy_tend_2nd / _4th (src/thermo_dry.cxx:557-562)
    std::string get_switch() const { return swthermo; }
    // Thermo_dry::exec (src/thermo_dry.cxx:551-565): the buoyancy tendency of w; defined after Fields/Grid below
    template<class G, class F> void exec(G& grid, F& fields, void* stream = nullptr)
    {
        if (swthermo != "dry") return;
        mhh_grid g = grid.abi();
        mhh_check(mhh_thermo_dry_buoyancy_tend(&g, swspatialorder, fields.mt.at("w")->fld_g, fields.sp.at(th)->fld_g, threfh_g, grav, stream));
    }
};
struct Stats {};                   // calc_tend is a no-op off sampling steps (src/stats.cxx:1893-1896)

template<typename TF>
inline mhh_fields abi_fields(const Fields<TF>& f, const Boundary<TF>* b = nullptr)
{
    mhh_fields a{};
    a.u = f.mp.at("u")->fld_g; a.v = f.mp.at("v")->fld_g; a.w = f.mp.at("w")->fld_g;
    a.ut = f.mt.at("u")->fld_g; a.vt = f.mt.at("v")->fld_g; a.wt = f.mt.at("w")->fld_g;
    int n = 0;
    for (auto& it : f.sp)
    {
        if (n >= MHH_MAX_SCALARS) throw std::runtime_error("mhh: more than MHH_MAX_SCALARS scalars");
        a.s[n] = it.second->fld_g; a.st[n] = f.st.at(it.first)->fld_g; a.svisc[n] = it.second->visc;
        a.s_fluxbot[n] = it.second->flux_bot_g; a.s_fluxtop[n] = it.second->flux_top_g;
        ++n;
    }
    a.nscalars = n;
    a.evisc = f.sd.count("evisc") ? f.sd.at("evisc")->fld_g : nullptr;
    a.p = f.sd.count("p") ? f.sd.at("p")->fld_g : nullptr;
    a.rhoref = f.rhoref_g; a.rhorefh = f.rhorefh_g; a.visc = f.visc;
    a.u_fluxbot = f.mp.at("u")->flux_bot_g; a.u_fluxtop = f.mp.at("u")->flux_top_g;
    a.v_fluxbot = f.mp.at("v")->flux_bot_g; a.v_fluxtop = f.mp.at("v")->flux_top_g;
    if (b) { a.dudz = b->dudz_g; a.dvdz = b->dvdz_g; a.dbdz = b->dbdz_g; a.z0m = b->z0m_g; }
    return a;
}
template<typename TF>
inline int scalar_index(const Fields<TF>& f, const std::string& name)
{
    int n = 0;
    for (auto& it : f.sp) { if (it.first == name) return n; ++n; }
    return -1;
}

// ---- Field3d_io (include/field3d_io.h; src/field3d_io.cxx:54-230) ---------------------------------------------------
// Restart files in the reference's layout: the interior (kmax x jtot x itot, C order) as a raw TF stream, data + offset.
// Host arrays; a y-slab rank (npy > 1) writes / reads its rows of the shared global file at j offset mpicoordy*jmax.
template<typename TF>
class Field3d_io
{
    public:
        explicit Field3d_io(Grid<TF>& gridin) : grid(gridin) {}
        // tmp1 holds imax*jmax*(kend-kstart) values; tmp2 is unused (the reference's transposed write needs it)
        int save_field3d(const TF* data, TF* tmp1, TF*, const char* filename, TF offset, int kstart, int kend)
        {
            const auto& gd = grid.get_grid_data();
            const int kmax = kend - kstart;
            for (int k=0; k<kmax; ++k)
                for (int j=0; j<gd.jmax; ++j)
                    for (int i=0; i<gd.imax; ++i)
                        tmp1[i + j*gd.imax + k*gd.imax*gd.jmax] = data[i+gd.igc + (j+gd.jgc)*gd.icells + (k+kstart)*gd.ijcells] + offset;
            FILE* f = std::fopen(filename, (gd.npy > 1 && gd.mpicoordy > 0) ? "r+b" : "wb");
            if (!f) return 1;
            int nerror = 0;
            for (int k=0; k<kmax && !nerror; ++k)
            {
                const long long pos = ((long long)k*gd.jtot + (long long)gd.mpicoordy*gd.jmax) * gd.itot * (long long)sizeof(TF);
                if (std::fseek(f, pos, SEEK_SET)) { ++nerror; break; }
                const size_t n = (size_t)gd.imax*gd.jmax;
                if (std::fwrite(tmp1 + (size_t)k*n, sizeof(TF), n, f) != n) ++nerror;
            }
            if (std::fclose(f)) ++nerror;
            return nerror;
        }
        int load_field3d(TF* data, TF* tmp1, TF*, const char* filename, TF offset, int kstart, int kend)
        {
            const auto& gd = grid.get_grid_data();
            const int kmax = kend - kstart;
            FILE* f = std::fopen(filename, "rb");
            if (!f) return 1;
            int nerror = 0;
            for (int k=0; k<kmax && !nerror; ++k)
            {
                const long long pos = ((long long)k*gd.jtot + (long long)gd.mpicoordy*gd.jmax) * gd.itot * (long long)sizeof(TF);
                if (std::fseek(f, pos, SEEK_SET)) { ++nerror; break; }
                const size_t n = (size_t)gd.imax*gd.jmax;
                if (std::fread(tmp1 + (size_t)k*n, sizeof(TF), n, f) != n) ++nerror;
            }
            std::fclose(f);
            if (nerror) return nerror;
            for (int k=0; k<kmax; ++k)
                for (int j=0; j<gd.jmax; ++j)
                    for (int i=0; i<gd.imax; ++i)
                        data[i+gd.igc + (j+gd.jgc)*gd.icells + (k+kstart)*gd.ijcells] = tmp1[i + j*gd.imax + k*gd.imax*gd.jmax] - offset;
            return 0;
        }
    private:
        Grid<TF>& grid;
};

// ---- Boundary_cyclic (include/boundary_cyclic.h:35-70) ----------------------------------------------------------
enum class Edge { East_west_edge, North_south_edge, Both_edges };
template<typename TF>
class Boundary_cyclic
{
    public:
        explicit Boundary_cyclic(Grid<TF>& gridin) : grid(gridin) {}
        void init() {}
        void exec_g(TF* data, void* stream = nullptr)    { mhh_grid g = grid.abi(); mhh_check(mhh_boundary_cyclic(&g, data, MHH_EDGE_BOTH, stream)); }
        void exec_g(TF* data, Edge e, void* stream = nullptr) { mhh_grid g = grid.abi(); mhh_check(mhh_boundary_cyclic(&g, data, static_cast<int>(e), stream)); }
        void exec_2d_g(TF* data, void* stream = nullptr) { mhh_grid g = grid.abi(); mhh_check(mhh_boundary_cyclic_2d(&g, data, stream)); }
        // the unsigned int overloads of include/boundary_cyclic.h:46-47 (index masks)
        void exec_g(unsigned int* data, Edge e = Edge::Both_edges, void* stream = nullptr) { mhh_grid g = grid.abi(); mhh_check(mhh_boundary_cyclic_u32(&g, data, static_cast<int>(e), stream)); }
        void exec_2d_g(unsigned int* data, void* stream = nullptr) { mhh_grid g = grid.abi(); mhh_check(mhh_boundary_cyclic_2d_u32(&g, data, stream)); }
    private:
        Grid<TF>& grid;
};

// ---- Advec (include/advec.h:45-70) ----------------------------------------------------------------------------
template<typename TF>
class Advec
{
    public:
        Advec(Grid<TF>& gridin, Fields<TF>& fieldsin, int schemein, double cflmaxin = 1.0, std::vector<std::string> fluxlimit_listin = {}) :
            fluxlimit_list(std::move(fluxlimit_listin)), grid(gridin), fields(fieldsin), scheme(schemein), cflmax(cflmaxin), cflmin(1.e-5), work(nullptr)
        {
            // src/advec_2i5.cxx:39-45: the limited scalars need a second vertical ghost level
            if (!fluxlimit_list.empty() && scheme != MHH_ADVEC_2I5 && scheme != MHH_ADVEC_2I62) throw std::runtime_error("fluxlimit_list is an option of swadvec=2i5 and 2i62");
            if (!fluxlimit_list.empty() && grid.get_grid_data().kgc < 2) throw std::runtime_error("fluxlimit_list needs kgc >= 2");
        }
        virtual ~Advec() {}
        // swadvec as in src/advec.cxx:55-83
        static std::shared_ptr<Advec> factory(Grid<TF>& g, Fields<TF>& f, const std::string& swadvec, double cflmax = 1.0,
                                              std::vector<std::string> fluxlimit_list = {})
        {
            int s;
            if (swadvec == "0") s = 0;                 // Advec_disabled (src/advec_disabled.cxx)
            else if (swadvec == "2") s = MHH_ADVEC_2; else if (swadvec == "2i5") s = MHH_ADVEC_2I5; else if (swadvec == "4") s = MHH_ADVEC_4;
            else if (swadvec == "2i4") s = MHH_ADVEC_2I4; else if (swadvec == "2i62") s = MHH_ADVEC_2I62; else if (swadvec == "2i53") s = MHH_ADVEC_2I53; else if (swadvec == "4m") s = MHH_ADVEC_4M;
            else throw std::runtime_error("\"" + swadvec + "\" is an illegal value for swadvec");
            return std::make_shared<Advec>(g, f, s, cflmax, std::move(fluxlimit_list));
        }
        void set_reduce_workspace(void* device_scratch) { work = device_scratch; }   // >= mhh_reduce_work_bytes()
        void create(Stats&) {}
        void exec(Stats&, void* stream = nullptr)
        {
            if (scheme == 0) return;
            mhh_grid g = grid.abi(); mhh_fields f = abi_fields(fields); mark_limited(f);
            mhh_check(mhh_advec_exec(&g, scheme, &f, stream));
        }
        // scalars named in advec.fluxlimit_list (src/advec_2i5.cxx:921)
        void mark_limited(mhh_fields& f) const
        {
            for (const std::string& nm : fluxlimit_list)
            {
                const int n = scalar_index(fields, nm);
                if (n < 0) throw std::runtime_error("fluxlimit_list: \"" + nm + "\" is not a prognostic scalar");
                f.s_fluxlimit[n] = 1;
            }
        }
        int get_scheme() const { return scheme; }
        std::vector<std::string> fluxlimit_list;
        double get_cfl(double dt, void* stream = nullptr)
        {
            if (scheme == 0) return cflmin;
            mhh_grid g = grid.abi(); double cfl = 0;
            mhh_check(mhh_advec_cfl(&g, scheme, fields.mp.at("u")->fld_g, fields.mp.at("v")->fld_g, fields.mp.at("w")->fld_g, dt, work, &cfl, stream));
            return cfl;
        }
        unsigned long get_time_limit(unsigned long idt, double dt, void* stream = nullptr)
        {
            if (scheme == 0) return ~0ul;              // Constants::ulhuge
            double cfl = get_cfl(dt, stream);
            cfl = std::max(cflmin, cfl);
            return idt * cflmax / cfl;
        }
    protected:
        Grid<TF>& grid; Fields<TF>& fields; int scheme; double cflmax; const double cflmin; void* work;
};

// ---- Diff (include/diff.h:37-71) -------------------------------------------------------------------------------
template<typename TF>
class Diff
{
    public:
        Diff(Grid<TF>& gridin, Fields<TF>& fieldsin, Boundary<TF>& boundaryin, int schemein, double dnmaxin = 0.4, TF csin = 0.23, TF tPrin = 1./3.) :
            tPr(tPrin), grid(gridin), fields(fieldsin), boundary(boundaryin), scheme(schemein), dnmax(dnmaxin), cs(csin), dnmul(0), mlen0_g(nullptr), work(nullptr), mlen2_g(nullptr), mlen2_neutral(false) {}
        virtual ~Diff() {}
        static std::shared_ptr<Diff> factory(Grid<TF>& g, Fields<TF>& f, Boundary<TF>& b, const std::string& swdiff, double dnmax = 0.4, TF cs = 0.23, TF tPr = 1./3.)
        {
            int s;
            if (swdiff == "0") s = 0;                  // Diff_disabled (src/diff_disabled.cxx)
            else if (swdiff == "2") s = MHH_DIFF_2; else if (swdiff == "4") s = MHH_DIFF_4; else if (swdiff == "smag2") s = MHH_DIFF_SMAG2;
            else throw std::runtime_error("\"" + swdiff + "\" is an illegal value for swdiff");
            return std::make_shared<Diff>(g, f, b, s, dnmax, cs, tPr);
        }
        void init() {}
        void set_reduce_workspace(void* device_scratch) { work = device_scratch; }
        // Diff_2/4::create: constant dnmul (src/diff_2.cxx:120-135); Diff_smag2 computes it per call
        void create(Stats&)
        {
            auto& gd = grid.get_grid_data();
            TF viscmax = fields.visc;
            for (auto& it : fields.sp) viscmax = std::max(it.second->visc, viscmax);
            dnmul = 0;
            for (int k=gd.kstart; k<gd.kend; ++k)
                dnmul = std::max(dnmul, std::abs(viscmax * (1./(gd.dx*gd.dx) + 1./(gd.dy*gd.dy) + 1./(gd.dz[k]*gd.dz[k]))));
        }
        // Diff_smag2::prepare_device (src/diff_smag2.cu:521-542): mlen0_device = caller-owned [kcells] device buffer that
        // receives the per-level mixing length; upload(dst_device, src_host, bytes) is the caller's H2D copy
        template<class Upload> void prepare_device(Boundary<TF>&, TF* mlen0_device, Upload upload)
        {
            if (scheme != MHH_DIFF_SMAG2) return;
            auto& gd = grid.get_grid_data();
            std::vector<TF> ml(gd.kcells);
            mhh_grid gh = grid.abi(true);
            mhh_check(mhh_smag2_mlen0_host(&gh, cs, ml.data()));
            upload(mlen0_device, ml.data(), ml.size()*sizeof(TF));
            mlen0_g = mlen0_device;
        }
        // Optional second table for a horizontally UNIFORM roughness length (swconstantz0-like set-ups): the squared mixing
        // length of calc_evisc per level (mhh_smag2_mlen2_host, same bits as the per-cell evaluation). Call after
        // prepare_device with the thermo switch known; mlen2_device = caller-owned [kcells] device buffer. Drop it
        // (clear_uniform_z0m) as soon as z0m may vary in the horizontal.
        template<class Upload> void set_uniform_z0m(TF z0m, bool neutral, TF* mlen2_device, Upload upload)
        {
            if (scheme != MHH_DIFF_SMAG2) return;
            auto& gd = grid.get_grid_data();
            std::vector<TF> ml(gd.kcells), m2(gd.kcells);
            mhh_grid gh = grid.abi(true);
            mhh_check(mhh_smag2_mlen0_host(&gh, cs, ml.data()));
            mhh_check(mhh_smag2_mlen2_host(&gh, boundary.get_switch() != "default", neutral ? 1 : 0, ml.data(), (double)z0m, m2.data()));
            upload(mlen2_device, m2.data(), m2.size()*sizeof(TF));
            mlen2_g = mlen2_device; mlen2_neutral = neutral;
        }
        void clear_uniform_z0m() { mlen2_g = nullptr; }
        void clear_device() { mlen0_g = nullptr; mlen2_g = nullptr; }
        void exec_viscosity(Thermo<TF>& thermo, void* stream = nullptr)
        {
            if (scheme != MHH_DIFF_SMAG2) return;
            mhh_grid g = grid.abi(); mhh_fields f = abi_fields(fields, &boundary); mhh_diff_params p = params(&thermo);
            mhh_check(mhh_diff_exec_viscosity(&g, scheme, &f, &p, stream));
        }
        // rows [j0, j1) (and, if j2 >= 0, [j2, j3) in the same launch) of exec_viscosity on a y-slab whose evisc ghost rows jstart-1
        // and jend are evaluated locally from the velocity halos instead of being exchanged (mhh_host_rccl.h, Substep_slab)
        void exec_viscosity_rows(Thermo<TF>& thermo, int j0, int j1, int j2 = -1, int j3 = -1, void* stream = nullptr)
        {
            if (scheme != MHH_DIFF_SMAG2) return;
            mhh_grid g = grid.abi(); mhh_fields f = abi_fields(fields, &boundary); mhh_diff_params p = params(&thermo);
            p.evisc_ghost_rows = 1;
            if (j2 < 0) mhh_check(mhh_diff_exec_viscosity_rows(&g, scheme, &f, &p, j0, j1, stream));
            else        mhh_check(mhh_diff_exec_viscosity_rows2(&g, scheme, &f, &p, j0, j1, j2, j3, stream));
        }
        // advec->exec + diff->exec on the rows [j0, j1) (and [j2, j3)): (advec_2i5, diff_smag2) only
        void exec_with_advec_rows(Advec<TF>& advec, int j0, int j1, int j2 = -1, int j3 = -1, void* stream = nullptr)
        {
            mhh_grid g = grid.abi(); mhh_fields f = abi_fields(fields, &boundary); mhh_diff_params p = params(nullptr);
            if (j2 < 0) mhh_check(mhh_rhs_exec_rows(&g, advec.get_scheme(), scheme, &f, &p, j0, j1, stream));
            else        mhh_check(mhh_rhs_exec_rows2(&g, advec.get_scheme(), scheme, &f, &p, j0, j1, j2, j3, stream));
        }
        void exec(Stats&, void* stream = nullptr)
        {
            if (scheme == 0) return;
            mhh_grid g = grid.abi(); mhh_fields f = abi_fields(fields, &boundary); mhh_diff_params p = params(nullptr);
            mhh_check(mhh_diff_exec(&g, scheme, &f, &p, stream));
        }
        int get_scheme() const { return scheme; }
        // advec->exec + diff->exec of Model::exec (src/model.cxx:388-392) as ONE pass over the fields where the library
        // has a fused kernel for the scheme pair: (2,2), (2i5,smag2). The 4th-order pair stays two calls because the
        // reference switches the w ghost cells between them (Boundary_w_type, src/model.cxx:387,389).
        // fold_buoyancy: also add Thermo_dry's buoyancy tendency (the caller then skips thermo->exec; nothing between it and
        // advec->exec touches wt, src/model.cxx:365-388)
        void exec_with_advec(Advec<TF>& advec, Stats& stats, void* stream = nullptr, Thermo<TF>* fold_buoyancy = nullptr)
        {
            const int a = advec.get_scheme();
            if (!((a == MHH_ADVEC_2 && scheme == MHH_DIFF_2) || (a == MHH_ADVEC_2I5 && scheme == MHH_DIFF_SMAG2)))
            {
                if (fold_buoyancy) fold_buoyancy->exec(grid, fields, stream);
                advec.exec(stats, stream); exec(stats, stream);
                return;
            }
            mhh_grid g = grid.abi(); mhh_fields f = abi_fields(fields, &boundary); mhh_diff_params p = params(nullptr);
            if (fold_buoyancy && fold_buoyancy->get_switch() == "dry")
            {
                p.buoyancy = fold_buoyancy->swspatialorder; p.th_for_N2 = scalar_index(fields, fold_buoyancy->th);
                p.threfh = fold_buoyancy->threfh_g; p.grav = fold_buoyancy->grav;
            }
            advec.mark_limited(f);
            mhh_check(mhh_rhs_exec(&g, a, scheme, &f, &p, stream));
        }
        double get_dn(double dt, void* stream = nullptr)
        {
            if (scheme == 0) return 1.e-9;                // Constants::dsmall (src/diff_disabled.cxx:56-60)
            if (scheme != MHH_DIFF_SMAG2) return dnmul*dt;
            mhh_grid g = grid.abi(); double d = 0;
            mhh_check(mhh_smag2_dnmul(&g, fields.sd.at("evisc")->fld_g, tPr, work, &d, stream));
            return d*dt;
        }
        unsigned long get_time_limit(unsigned long idt, double dt, void* stream = nullptr)
        {
            if (scheme == 0) return ~0ul;                 // Constants::ulhuge
            if (scheme != MHH_DIFF_SMAG2) return idt * dnmax / (dt * dnmul);
            mhh_grid g = grid.abi(); double d = 0;
            mhh_check(mhh_smag2_dnmul(&g, fields.sd.at("evisc")->fld_g, tPr, work, &d, stream));
            d = std::max(1.e-9, d);                       // Constants::dsmall
            return idt * dnmax / (dt * d);
        }
        mhh_diff_params params(Thermo<TF>* thermo) const
        {
            mhh_diff_params p{};
            p.cs = cs; p.tPr = tPr; p.surface_model = (boundary.get_switch() != "default"); p.mlen0 = mlen0_g;
            p.neutral = thermo ? (thermo->get_switch() == "0") : 0;
            if (mlen2_g && mlen2_neutral == (p.neutral != 0)) p.mlen2 = mlen2_g;
            p.th_for_N2 = -1;
            if (thermo && !p.neutral)
            {
                p.N2 = thermo->N2_g;
                if (!p.N2) { p.th_for_N2 = scalar_index(fields, thermo->th); p.thref = thermo->thref_g; p.grav = thermo->grav; }
            }
            return p;
        }
        TF tPr;
    protected:
        Grid<TF>& grid; Fields<TF>& fields; Boundary<TF>& boundary; int scheme; double dnmax; TF cs; double dnmul; TF* mlen0_g; void* work; TF* mlen2_g; bool mlen2_neutral;
};

// ---- Pres (include/pres.h:39-85) ---------------------------------------------------------------------------------
template<typename TF>
class Pres
{
    public:
        Pres(Grid<TF>& gridin, Fields<TF>& fieldsin, int orderin) : grid(gridin), fields(fieldsin), order(orderin), plan(nullptr), work(nullptr) {}
        virtual ~Pres() { clear_device(); }
        static std::shared_ptr<Pres> factory(Grid<TF>& g, Fields<TF>& f, const std::string& swpres)
        {
            if (swpres == "0") return std::make_shared<Pres>(g, f, 0);     // Pres_disabled (src/pres_disabled.cxx): every member does nothing
            if (swpres == "2") return std::make_shared<Pres>(g, f, 2);
            if (swpres == "4") return std::make_shared<Pres>(g, f, 4);
            throw std::runtime_error("\"" + swpres + "\" is an illegal value for swpres");
        }
        void init() {}
        void set_values() {}                  // the coefficient tables are built inside prepare_device from the host metrics
        void create(Stats&) {}
        void set_reduce_workspace(void* device_scratch) { work = device_scratch; }
        void prepare_device()
        {
            clear_device();
            if (order == 0) return;
            auto& gd = grid.get_grid_data();
            mhh_grid gh = grid.abi(true);
            mhh_check(mhh_pres_plan_create(&gh, order, gd.dz.data(), gd.dzhi.data(), gd.dzi4.data(), gd.dzhi4.data(), fields.rhoref.data(), fields.rhorefh.data(), &plan));
        }
        void clear_device() { if (plan) { mhh_pres_plan_destroy(plan); plan = nullptr; } }
        void exec(double dt, Stats&, void* stream = nullptr)
        {
            if (order == 0) return;
            if (!plan) throw std::runtime_error("Pres::exec before prepare_device");
            mhh_grid g = grid.abi(); mhh_fields f = abi_fields(fields);
            mhh_check(mhh_pres_exec(plan, &g, &f, dt, stream));
        }
        // pres->exec(sub_dt) and the momentum part of timeloop.exec() (src/model.cxx:411,484; src/timeloop.cxx:250-334) in one call:
        // the sub-step of u, v, w is applied in the kernel that stores the corrected tendencies. The caller's Timeloop::exec then
        // covers the scalars only. Same bits as exec() followed by the three sub-steps.
        void exec_with_timeloop(double sub_dt, int rkorder, int substep, double dt, Stats&, void* stream = nullptr)
        {
            if (order == 0) return;
            if (!plan) throw std::runtime_error("Pres::exec before prepare_device");
            mhh_grid g = grid.abi(); mhh_fields f = abi_fields(fields);
            mhh_check(mhh_pres_exec_rk(plan, &g, &f, sub_dt, rkorder, substep, dt, stream));
        }
        TF check_divergence(void* stream = nullptr)
        {
            if (order == 0) return TF(0);                          // src/pres_disabled.cxx: check_divergence returns 0
            mhh_grid g = grid.abi(); mhh_fields f = abi_fields(fields); double d = 0;
            mhh_check(mhh_pres_check_divergence(&g, order, &f, work, &d, stream));
            return static_cast<TF>(d);
        }
    protected:
        Grid<TF>& grid; Fields<TF>& fields; int order; mhh_pres_plan* plan; void* work;
};

} // namespace mhh_host
