// Empty stand-in for Jittor's utils/log.h (external framework header, not part of /root/reference).
#pragma once
