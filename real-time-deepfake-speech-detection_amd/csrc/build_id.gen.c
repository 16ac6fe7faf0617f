const char afx_build_id_str[] = "d9b3c4711839";
