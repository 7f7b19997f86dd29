// fast_limo/Modules/Mapper.hpp -- GPU-resident map behind the reference's Mapper API
// (reference Modules/Mapper.hpp:48-71, Modules/Mapper.cpp:38-96).  The octree is replaced by the
// uniform-grid index of libflimo_hip; `match` keeps its signature for API compatibility, the
// filter itself uses the reduced seam (match_reduce) instead of materialising Matches.
#ifndef __FASTLIMO_MAPPER_HPP__
#define __FASTLIMO_MAPPER_HPP__
#include <atomic>
#include <condition_variable>
#include <mutex>
#include <thread>
#include "fast_limo/Common.hpp"
#include "fast_limo/Objects/Match.hpp"
#include "fast_limo/Objects/State.hpp"
#include "fast_limo/Utils/Config.hpp"

struct flimo_ctx;

class fast_limo::Mapper {
 public:
  Matches matches;

  Mapper();
  ~Mapper();
  void set_num_threads(int n);
  void set_config(const Config::iKFoM::Mapping& cfg);
  bool exists();
  int size();
  double last_time();
  // Mapper::match (Mapper.cpp:59-86): matches of the RESIDENT scan (pc must be the cloud last
  // handed to the Localizer / set with set_scan) at state s.
  Matches match(State s, pcl::PointCloud<PointType>::Ptr& pc);
  void add(pcl::PointCloud<PointType>::Ptr& pc, double time);

  // --- MI355X additions -----------------------------------------------------------------------
  // one Mapper per GPU; getInstance() keeps the reference's process-wide singleton on device 0
  explicit Mapper(int device);
  bool attach(int device, float cell_size);     // creates the flimo_ctx; false + message on failure
  // The handle of the GPU context.  A map insert started by add_scan() may still be running on the Mapper's worker
  // thread: ctx() (like every other method of this class) waits for it first, so whoever holds the handle sees a
  // quiescent context.  Re-fetch it after each Localizer::updatePointCloud instead of caching it across scans.
  flimo_ctx* ctx() { sync(); return ctx_; }
  // A second context on the same GPU for the INPUT stage of a sweep (upload, filters, stamps, time order: nothing of it reads the
  // map): handed out WITHOUT waiting for a running insert, so that stage overlaps the previous sweep's Mapper::add; the sweep is
  // then handed over to ctx() with flimo_scan_adopt.  nullptr when it cannot be created (the caller uses ctx()).
  flimo_ctx* front_ctx();
  // Path exit of a scan (reference Localizer.cpp:361-377: transformPointCloud + Mapper::add) for the scan RESIDENT on
  // the GPU: returns at once, the insert runs on the worker thread and overlaps the host-side preparation (filters,
  // time sort) of the next scan.  FLIMO_SYNC_INSERT=1 (or set_async(false)) makes it synchronous.
  void add_scan(const double x26[26], double stamp);
  void sync();                                  // wait for a running insert (no-op when idle)
  void set_async(bool on) { sync(); async_ = on; }
  double last_insert_seconds() { sync(); return insert_seconds_; }
  double last_handoff_time() const { return handoff_time_; }      // developer timing
  const Config::iKFoM::Mapping& config_ref() const { return config; }
  const std::string& last_error() const { return err_; }

  static Mapper& getInstance() {
    static Mapper* mapper = new Mapper();
    return *mapper;
  }

 private:
  Config::iKFoM::Mapping config;
  int num_threads_;
  flimo_ctx* ctx_;
  flimo_ctx* front_;            // the input stage's context (front_ctx())
  int device_;
  float cell_size_;
  std::string err_;
  // worker of add_scan()
  bool async_;
  std::thread worker_;
  std::mutex wm_;
  std::condition_variable wcv_;
  std::atomic<bool> busy_{false}, quit_{false};   // written under wm_, also polled without it (short spins before the condition-variable waits)
  double job_x_[26];
  double job_stamp_ = 0.0;
  double handoff_time_ = 0.0;
  double insert_seconds_ = 0.0;
  void run_insert(const double x26[26], double stamp);
  void worker_main();
  Mapper(const Mapper&) = delete;
  Mapper& operator=(const Mapper&) = delete;
};
#endif
