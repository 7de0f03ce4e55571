// include/xwb_endpoint.hpp -- the reference's localhost TCP RPC, spoken by one env slot of a batch.
//
// In the reference every environment lives in a child process as a `SimulatorClient`, and the trainer holds one
// `SimulatorServer` per environment that calls it remotely (simulator_interface.h:169-264, simulator_interface.cpp:
// 170-435, simulator_communication.{h,cpp}).  This header restates that protocol over plain POSIX sockets (the
// reference uses Boost.Asio) so that an unmodified reference-side trainer can drive ONE SLOT of a batched simulator:
//
//   xwb::SlotClient      = SimulatorClient: connects to the trainer's port (5 attempts, 1 s apart), sends its game
//                          name, expects "accepted", then serves "reset" | "take_actions" | "get_state" |
//                          "report_perf" | "get_extra_info" until "stop".
//   xwb::RemoteSimulator = SimulatorServer: the trainer's side (listens on an ephemeral port, accepts one client,
//                          checks the greeting, remote-calls the verbs).  Shipped because the in-process batch can
//                          equally be driven THROUGH the reference's wire by C++ code that holds a SimInterfacePtr;
//                          the tests use it as the stand-in for the reference trainer.
//
// Wire (simulator_communication.h:34-77,cpp:31-48; memory_util.h:304-333): message = size_t body size (8 bytes,
// host order) + body; body fields are appended in call order: PODs raw, std::string = size_t length + bytes + NUL,
// a StatePacket (if any) LAST in its own encoding (compose_msg(sim_data, args...) appends args first).
//   reset        ->  "reset", int num_actions, int game_over, int lives, size_t h, w, c, double X, Y, Z
//   take_actions <-  "take_actions", int act_rep, bool show_screen, StatePacket actions
//                ->  "take_actions", float reward, int64 num_steps, int game_over, int lives, bool success, string last
//   get_state    <-  "get_state", float reward           ->  "get_state", StatePacket state
//   get_extra_info -> "get_extra_info", string info ;  report_perf -> (the unchanged request body, as the reference
//   does: the client never re-composes the message for this verb) ;  stop: no reply.
// Parity: StatePacket bytes are pinned by the reference's tests/test_statepacket.cpp; the message framing has no
// reference test and is restated from the sources cited above ("parity unpinned" for the framing).
#pragma once

#include "xwb_simulator.hpp"

#include <arpa/inet.h>
#include <netinet/in.h>
#include <netinet/tcp.h>
#include <sys/socket.h>
#include <unistd.h>

namespace xwb {

namespace wire {

// util::BinaryBuffer as far as the RPC uses it: append at the end, read from a cursor
class Message {
  public:
    void clear() { b_.clear(); at_ = 0; }
    void rewind() { at_ = 0; }
    size_t size() const { return b_.size(); }
    const uint8_t *data() const { return b_.data(); }
    void resize(size_t n) { b_.resize(n); }
    uint8_t *data_mutable() { return b_.data(); }

    template <typename T>
    void append(const T &t) { const uint8_t *p = reinterpret_cast<const uint8_t *>(&t); b_.insert(b_.end(), p, p + sizeof(T)); }
    void append(const std::string &s) {                      // memory_util.h:311-314
        append((size_t)s.length());
        b_.insert(b_.end(), s.c_str(), s.c_str() + s.length() + 1);
    }
    void append(const char *s) { append(std::string(s)); }
    template <typename T>
    void append(const std::vector<T> &v) {                   // memory_util.h:297-304: element count, then the elements
        append((size_t)v.size());
        append(v.data(), v.size());
    }
    template <typename T>
    void append(const T *a, size_t n) {                      // memory_util.h:285-291: raw elements, no count
        const uint8_t *p = reinterpret_cast<const uint8_t *>(a);
        b_.insert(b_.end(), p, p + n * sizeof(T));
    }
    void append(const Message &m) { b_.insert(b_.end(), m.b_.begin(), m.b_.end()); }   // memory_util.h:316-319
    template <typename T>
    void insert(size_t offset, const T *a, size_t n) {       // memory_util.h:321-333
        if (offset > b_.size()) throw Error("wire::Message: insert beyond the end");
        const uint8_t *p = reinterpret_cast<const uint8_t *>(a);
        b_.insert(b_.begin() + (std::ptrdiff_t)offset, p, p + n * sizeof(T));
    }
    bool eof() const { return at_ >= b_.size(); }
    void append(const StatePacket &p) { const std::vector<uint8_t> e = p.encode(); b_.insert(b_.end(), e.begin(), e.end()); }

    template <typename T>
    void read(T &t) { need(sizeof(T)); memcpy(&t, b_.data() + at_, sizeof(T)); at_ += sizeof(T); }
    void read(std::string &s) {
        size_t n; read(n);
        if (n >= b_.size() - at_) throw Error("wire::Message: truncated message");      // n + 1 bytes, without overflow
        s.assign(reinterpret_cast<const char *>(b_.data() + at_), n);
        at_ += n + 1;
    }
    template <typename T>
    void read(std::vector<T> &v) {                           // memory_util.h:362-369
        size_t n; read(n);
        if (n > (b_.size() - at_) / sizeof(T)) throw Error("wire::Message: truncated message");   // before sizing anything by it
        v.resize(n);
        read(v.data(), n);
    }
    template <typename T>
    void read(T *a, size_t n) {
        if (n > (b_.size() - at_) / sizeof(T)) throw Error("wire::Message: truncated message");
        if (n) memcpy(a, b_.data() + at_, n * sizeof(T));
        at_ += n * sizeof(T);
    }
    void read(StatePacket &p) { p.decode(b_.data() + at_, b_.size() - at_); at_ = b_.size(); }   // always last

  private:
    void need(size_t n) const { if (at_ > b_.size() || n > b_.size() - at_) throw Error("wire::Message: truncated message"); }
    std::vector<uint8_t> b_;
    size_t at_ = 0;
};

constexpr size_t kMaxMessageBytes = size_t(256) << 20;

class Socket {
  public:
    Socket() {}
    ~Socket() { close(); }
    Socket(const Socket &) = delete;
    Socket &operator=(const Socket &) = delete;

    // CommServer: acceptor on an ephemeral IPv4 port (simulator_communication.cpp:51-60)
    int listen_any() {
        lfd_ = ::socket(AF_INET, SOCK_STREAM, 0);
        if (lfd_ < 0) throw Error("socket() failed");
        sockaddr_in a{};
        // the reference's acceptor listens on every interface; its clients only ever connect to "localhost"
        // (simulator_communication.cpp:63-91) and the protocol has no authentication: loopback only
        a.sin_family = AF_INET; a.sin_addr.s_addr = htonl(INADDR_LOOPBACK); a.sin_port = 0;
        socklen_t len = sizeof a;
        if (::bind(lfd_, reinterpret_cast<sockaddr *>(&a), sizeof a) != 0 || ::listen(lfd_, 1) != 0 ||
            ::getsockname(lfd_, reinterpret_cast<sockaddr *>(&a), &len) != 0)
            throw Error("bind/listen failed");
        return ntohs(a.sin_port);
    }
    bool accept_one() {
        fd_ = ::accept(lfd_, nullptr, nullptr);
        nodelay();
        return fd_ >= 0;
    }
    // CommClient::establish_connection: "localhost", MAX_ATTEMPTS = 5, sleep(1) between (cpp:63-91)
    bool connect_local(int port, int attempts = 5) {
        for (int i = 0; i < attempts; ++i) {
            fd_ = ::socket(AF_INET, SOCK_STREAM, 0);
            if (fd_ < 0) throw Error("socket() failed");
            sockaddr_in a{};
            a.sin_family = AF_INET; a.sin_addr.s_addr = htonl(INADDR_LOOPBACK); a.sin_port = htons((uint16_t)port);
            if (::connect(fd_, reinterpret_cast<sockaddr *>(&a), sizeof a) == 0) { nodelay(); return true; }
            ::close(fd_);
            fd_ = -1;
            ::sleep(1);
        }
        return false;
    }
    void close() {
        if (fd_ >= 0) { ::shutdown(fd_, SHUT_RDWR); ::close(fd_); fd_ = -1; }
        if (lfd_ >= 0) { ::close(lfd_); lfd_ = -1; }
    }
    // Communicator::deliver_msg / receive_msg (cpp:31-48)
    void deliver(const Message &m) {
        const size_t n = m.size();
        send_all(&n, sizeof n);
        send_all(m.data(), n);
    }
    void receive(Message &m) {
        size_t n = 0;
        recv_all(&n, sizeof n);
        // the largest legitimate message is a state packet: a context ring of float frames (16 frames of 3 x 256 x 256
        // floats = 12.6 MB); 256 MiB is far beyond it and far below what a hostile length could make us allocate
        if (n > kMaxMessageBytes) throw Error("wire: implausible message size");
        m.clear();
        m.resize(n);
        recv_all(m.data_mutable(), n);
        m.rewind();
    }

  private:
    void nodelay() { int one = 1; if (fd_ >= 0) ::setsockopt(fd_, IPPROTO_TCP, TCP_NODELAY, &one, sizeof one); }
    void send_all(const void *p, size_t n) {
        const char *c = static_cast<const char *>(p);
        while (n) {
            const ssize_t k = ::send(fd_, c, n, MSG_NOSIGNAL);
            if (k <= 0) throw Error("wire: send failed");
            c += k; n -= (size_t)k;
        }
    }
    void recv_all(void *p, size_t n) {
        char *c = static_cast<char *>(p);
        while (n) {
            const ssize_t k = ::recv(fd_, c, n, 0);
            if (k <= 0) throw Error("wire: connection closed");
            c += k; n -= (size_t)k;
        }
    }
    int fd_ = -1, lfd_ = -1;
};

}  // namespace wire

// --------------------------------------------------------------------- SlotClient ----
// SimulatorClient (simulator_interface.cpp:316-435) for one env slot.
class SlotClient {
  public:
    SlotClient(std::shared_ptr<BatchedSimulator> batch, int env, const std::string &name, int port)
        : game_(std::move(batch), env), name_(name), port_(port) {}

    // establish_connection + simulation_loop; returns false when the connection or the greeting failed
    bool start() {
        if (!sock_.connect_local(port_)) return false;
        wire::Message m;
        m.append(name_);
        sock_.deliver(m);
        sock_.receive(m);
        std::string reply;
        m.read(reply);
        if (reply != "accepted") { sock_.close(); return false; }
        game_.start();
        loop();
        sock_.close();
        game_.stop();
        return true;
    }

  private:
    void loop() {
        wire::Message m;
        std::string cmd;
        while (true) {
            sock_.receive(m);
            m.read(cmd);
            if (cmd == "reset") {
                game_.reset_game();
                size_t h, w, c;
                double X, Y, Z;
                game_.get_screen_out_dimensions(h, w, c);
                game_.get_world_dimensions(X, Y, Z);
                const int na = game_.get_num_actions(), over = game_.game_over(), lives = game_.get_lives();
                m.clear();
                m.append("reset"); m.append(na); m.append(over); m.append(lives);
                m.append(h); m.append(w); m.append(c); m.append(X); m.append(Y); m.append(Z);
            } else if (cmd == "take_actions") {
                int act_rep; bool show; StatePacket actions;
                m.read(act_rep); m.read(show); m.read(actions);
                const float r = game_.take_actions(actions, act_rep, show);
                const int64_t steps = game_.get_num_steps();
                const int over = game_.game_over(), lives = game_.get_lives();
                const bool ok = game_.last_action_success();
                m.clear();
                m.append("take_actions"); m.append(r); m.append(steps); m.append(over); m.append(lives); m.append(ok);
                m.append(game_.last_action());
            } else if (cmd == "get_state") {
                float reward;
                m.read(reward);
                const StatePacket st = game_.get_state(reward);
                m.clear();
                m.append("get_state"); m.append(st);
            } else if (cmd == "get_extra_info") {
                std::string info;
                game_.get_extra_info(info);
                m.clear();
                m.append("get_extra_info"); m.append(info);
            } else if (cmd == "report_perf") {
                // teacher_report_task_performance() (simulator_interface.cpp:372-373): the table is logged on this side, as in
                // the reference; the request body goes back unchanged
                game_.teacher_report_task_performance();
            } else if (cmd == "stop") {
                break;
            }
            sock_.deliver(m);
        }
    }

    SimulatorInterface game_;
    std::string name_;
    int port_;
    wire::Socket sock_;
};

// ---------------------------------------------------------------- RemoteSimulator ----
// SimulatorServer (simulator_interface.cpp:170-313): the trainer-side proxy of one remote environment.
class RemoteSimulator {
  public:
    explicit RemoteSimulator(const std::string &name) : name_(name) { port_ = sock_.listen_any(); }
    int port() const { return port_; }

    bool start() {                                            // establish_connection + greeting check
        if (!sock_.accept_one()) return false;
        wire::Message m;
        sock_.receive(m);
        std::string greeting;
        m.read(greeting);
        if (greeting != name_) return false;                  // reference: CHECK_EQ
        m.clear();
        m.append("accepted");
        sock_.deliver(m);
        return true;
    }
    void stop() {
        wire::Message m;
        m.append("stop");
        sock_.deliver(m);
        sock_.close();
    }
    void reset_game() {
        wire::Message m = call("reset", nullptr);
        m.read(num_actions_); m.read(game_over_code_); m.read(lives_);
        m.read(height_); m.read(width_); m.read(channels_); m.read(X_); m.read(Y_); m.read(Z_);
        num_steps_ = 0;
    }
    float take_actions(const StatePacket &actions, int act_rep, bool show_screen) {
        wire::Message req;
        req.append("take_actions"); req.append(act_rep); req.append(show_screen); req.append(actions);
        wire::Message m = roundtrip(req, "take_actions");
        float r; int64_t steps;
        num_steps_++;
        m.read(r); m.read(steps); m.read(game_over_code_); m.read(lives_); m.read(last_action_success_); m.read(last_action_);
        if (steps != num_steps_) throw Error("RemoteSimulator: num_steps out of sync");   // reference: CHECK_EQ
        return r;
    }
    StatePacket get_state(float reward) {
        wire::Message req;
        req.append("get_state"); req.append(reward);
        wire::Message m = roundtrip(req, "get_state");
        StatePacket st;
        m.read(st);
        return st;
    }
    void get_extra_info(std::string &info) { wire::Message m = call("get_extra_info", nullptr); m.read(info); }
    void teacher_report_task_performance() { call("report_perf", nullptr); }

    int game_over() const { return game_over_code_; }
    int get_num_actions() const { return num_actions_; }
    int get_lives() const { return lives_; }
    int64_t get_num_steps() const { return num_steps_; }
    void get_screen_out_dimensions(size_t &h, size_t &w, size_t &c) const { h = height_; w = width_; c = channels_; }
    void get_world_dimensions(double &X, double &Y, double &Z) const { X = X_; Y = Y_; Z = Z_; }
    bool last_action_success() const { return last_action_success_; }
    std::string last_action() const { return last_action_; }

  private:
    wire::Message call(const char *func, const StatePacket *) {
        wire::Message req;
        req.append(func);
        return roundtrip(req, func);
    }
    wire::Message roundtrip(const wire::Message &req, const char *func) {    // call_remote_func (h:219-235)
        sock_.deliver(req);
        wire::Message m;
        sock_.receive(m);
        std::string reply;
        m.read(reply);
        if (reply != func) throw Error(std::string("RemoteSimulator: unexpected reply to ") + func);
        return m;
    }

    std::string name_;
    int port_ = 0;
    wire::Socket sock_;
    int num_actions_ = -1, game_over_code_ = 0, lives_ = 0;
    int64_t num_steps_ = -1;
    size_t height_ = 0, width_ = 0, channels_ = 0;
    double X_ = 0, Y_ = 0, Z_ = 0;
    bool last_action_success_ = false;
    std::string last_action_;
};

}  // namespace xwb
